// oracle/shim/g2o/core/optimizable_graph.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  g2o is an un-vendored dependency of
// the reference (g2o/Makefile:5-8) and not installed here.  The reference's LocalMap (dvo_slam/src/local_map.cpp) uses it as the
// container of its frame vertices and relative-pose edges; graph OPTIMISATION is outside this engine's scope, so the stand-in
// stores vertices / edges and aborts on optimize().
#pragma once

#include <cstdlib>
#include <iostream>
#include <set>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

namespace g2o {

class OptimizableGraph {
 public:
  class Data {
   public:
    virtual ~Data() {}
    virtual bool read(std::istream&) = 0;
    virtual bool write(std::ostream&) const = 0;
  };
  class Edge;
  typedef std::set<Edge*> EdgeSet;
  class Vertex {
   public:
    Vertex() : id_(-1), fixed_(false), data_(0) {}
    virtual ~Vertex() { delete data_; }
    void setId(int id) { id_ = id; }
    int id() const { return id_; }
    void setFixed(bool f) { fixed_ = f; }
    bool fixed() const { return fixed_; }
    void setUserData(Data* d) { data_ = d; }
    Data* userData() const { return data_; }
    EdgeSet& edges() { return edges_; }
   private:
    int id_;
    bool fixed_;
    Data* data_;
    EdgeSet edges_;
  };
  class Edge {
   public:
    Edge() : id_(-1) {}
    virtual ~Edge() {}
    void setId(int id) { id_ = id; }
    int id() const { return id_; }
    void resize(size_t n) { vertices_.resize(n); }
    void setVertex(size_t i, Vertex* v) { vertices_[i] = v; }
    Vertex* vertex(size_t i) const { return vertices_[i]; }
    const std::vector<Vertex*>& vertices() const { return vertices_; }
   private:
    int id_;
    std::vector<Vertex*> vertices_;
  };
  virtual ~OptimizableGraph() {
    for (size_t i = 0; i < edges_.size(); ++i) delete edges_[i];
    for (size_t i = 0; i < vertices_.size(); ++i) delete vertices_[i];
  }
  bool addVertex(Vertex* v) { vertices_.push_back(v); return true; }
  bool addEdge(Edge* e) {
    edges_.push_back(e);
    for (size_t i = 0; i < e->vertices().size(); ++i) e->vertex(i)->edges().insert(e);
    return true;
  }
 protected:
  std::vector<Vertex*> vertices_;
  std::vector<Edge*> edges_;
};

}  // namespace g2o
