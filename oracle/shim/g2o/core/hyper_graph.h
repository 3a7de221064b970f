// oracle/shim/g2o/core/hyper_graph.h -- TEST INFRASTRUCTURE, see oracle/shim/Eigen/Core.  g2o is an un-vendored dependency of the
// reference (g2o/Makefile:5-8, HEAD, unpinned) and not installed here.  The reference's LocalMap (dvo_slam/src/local_map.cpp) and
// KeyframeGraph (dvo_slam/src/keyframe_graph.cpp) use it as the CONTAINER of their pose vertices and relative-pose edges -- ids,
// incidence sets, user data, levels, robust kernels -- and as the optimiser.  The container half is restated here with g2o's published
// semantics (vertices keyed by id, an edge registered with each of its vertices, removeEdge / changeId as in hyper_graph.cpp); the
// optimiser half is NOT: SparseOptimizer::optimize leaves every estimate where it is (sparse_optimizer.h).
#pragma once

#include <cstddef>
#include <map>
#include <set>
#include <vector>

namespace g2o {

class HyperGraph {
 public:
  class Vertex;
  class Edge;
  typedef std::set<Edge*> EdgeSet;
  typedef std::set<Vertex*> VertexSet;
  typedef std::map<int, Vertex*> VertexIDMap;            // (g2o: an unordered map; no caller here depends on its order)
  typedef std::vector<Vertex*> VertexContainer;

  class Vertex {
   public:
    explicit Vertex(int id = -1) : id_(id) {}
    virtual ~Vertex() {}
    int id() const { return id_; }
    virtual void setId(int id) { id_ = id; }
    const EdgeSet& edges() const { return edges_; }
    EdgeSet& edges() { return edges_; }
   private:
    int id_;
    EdgeSet edges_;
  };

  class Edge {
   public:
    explicit Edge(int id = -1) : id_(id) {}
    virtual ~Edge() {}
    virtual void resize(size_t n) { vertices_.resize(n, 0); }
    const VertexContainer& vertices() const { return vertices_; }
    VertexContainer& vertices() { return vertices_; }
    const Vertex* vertex(size_t i) const { return vertices_[i]; }
    Vertex* vertex(size_t i) { return vertices_[i]; }
    void setVertex(size_t i, Vertex* v) { vertices_[i] = v; }
    int id() const { return id_; }
    void setId(int id) { id_ = id; }
   private:
    int id_;
    VertexContainer vertices_;
  };

  HyperGraph() {}
  virtual ~HyperGraph() { clear(); }
  Vertex* vertex(int id) {
    VertexIDMap::iterator it = vertices_.find(id);
    return it == vertices_.end() ? 0 : it->second;
  }
  const Vertex* vertex(int id) const {
    VertexIDMap::const_iterator it = vertices_.find(id);
    return it == vertices_.end() ? 0 : it->second;
  }
  const VertexIDMap& vertices() const { return vertices_; }
  VertexIDMap& vertices() { return vertices_; }
  const EdgeSet& edges() const { return edges_; }
  EdgeSet& edges() { return edges_; }

  virtual bool addVertex(Vertex* v) {
    if (vertices_.count(v->id())) return false;
    vertices_[v->id()] = v;
    return true;
  }
  virtual bool addEdge(Edge* e) {
    if (!edges_.insert(e).second) return false;
    for (size_t i = 0; i < e->vertices().size(); ++i)
      if (e->vertex(i)) e->vertex(i)->edges().insert(e);
    return true;
  }
  virtual bool removeEdge(Edge* e) {                      // detaches and deletes, like g2o
    EdgeSet::iterator it = edges_.find(e);
    if (it == edges_.end()) return false;
    edges_.erase(it);
    for (size_t i = 0; i < e->vertices().size(); ++i)
      if (e->vertex(i)) e->vertex(i)->edges().erase(e);
    delete e;
    return true;
  }
  virtual bool changeId(Vertex* v, int new_id) {
    Vertex* known = vertex(v->id());
    if (known != v) return false;
    vertices_.erase(v->id());
    v->setId(new_id);
    vertices_[new_id] = v;
    return true;
  }
  virtual void clear() {
    for (EdgeSet::iterator it = edges_.begin(); it != edges_.end(); ++it) delete *it;
    for (VertexIDMap::iterator it = vertices_.begin(); it != vertices_.end(); ++it) delete it->second;
    edges_.clear();
    vertices_.clear();
  }
 private:
  HyperGraph(const HyperGraph&);
  HyperGraph& operator=(const HyperGraph&);
  VertexIDMap vertices_;
  EdgeSet edges_;
};

}  // namespace g2o
