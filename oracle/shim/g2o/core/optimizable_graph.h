// oracle/shim/g2o/core/optimizable_graph.h -- TEST INFRASTRUCTURE, see hyper_graph.h.
#pragma once

#include <cstdlib>
#include <iostream>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include "hyper_graph.h"
#include "robust_kernel.h"

namespace g2o {

class OptimizableGraph : public HyperGraph {
 public:
  class Data {
   public:
    virtual ~Data() {}
    virtual bool read(std::istream&) = 0;
    virtual bool write(std::ostream&) const = 0;
  };
  class Vertex : public HyperGraph::Vertex {
   public:
    Vertex() : fixed_(false), marginalized_(false), hessian_index_(-1), data_(0) {}
    virtual ~Vertex() { delete data_; }
    void setFixed(bool f) { fixed_ = f; }
    bool fixed() const { return fixed_; }
    void setMarginalized(bool m) { marginalized_ = m; }
    bool marginalized() const { return marginalized_; }
    void setHessianIndex(int i) { hessian_index_ = i; }
    int hessianIndex() const { return hessian_index_; }
    void setUserData(Data* d) { data_ = d; }               // (the vertex owns it; callers hand it over with setUserData(0) on the donor)
    Data* userData() const { return data_; }
   private:
    bool fixed_, marginalized_;
    int hessian_index_;
    Data* data_;
  };
  class Edge : public HyperGraph::Edge {
   public:
    Edge() : level_(0), kernel_(0) {}
    virtual ~Edge() { delete kernel_; }
    int level() const { return level_; }
    void setLevel(int l) { level_ = l; }
    RobustKernel* robustKernel() const { return kernel_; }
    void setRobustKernel(RobustKernel* k) { delete kernel_; kernel_ = k; }
    virtual double chi2() const { return 0.0; }
   private:
    int level_;
    RobustKernel* kernel_;
  };
  Vertex* vertex(int id) { return static_cast<Vertex*>(HyperGraph::vertex(id)); }
  const Vertex* vertex(int id) const { return static_cast<const Vertex*>(HyperGraph::vertex(id)); }
};

}  // namespace g2o
