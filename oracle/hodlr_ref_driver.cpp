// TEST INFRASTRUCTURE ONLY -- builds into oracle/_ref/_hodlr*.so (oracle/Makefile), never linked
// into libgeorge_amd.so.
//
// A Python module with the interface of the reference's `george.solvers._hodlr`
// (/root/reference/src/george/solvers/_hodlr.cpp:112-204), around the reference's UNMODIFIED
// `george/hodlr.h` -- compiled where it lies against oracle/mini_eigen (the stand-in for the absent
// Eigen submodule) -- and the reference's own kernel tree (`george/kernels.h`, `george/parser.h`).
// `_hodlr.cpp` itself cannot be used: it needs pybind11/eigen.h, i.e. the real Eigen.  This driver
// follows it line by line instead: SolverMatrix (_hodlr.cpp:13-36), Solver::compute (:55-94: parse
// spec, mt19937 seeded with `seed`, diag = yerr^2, Node(diag, matrix, 0, n, min_size, tol, random),
// compute(), log_determinant()), apply_inverse / dot_solve / get_inverse (:156-203).
//
// One extension for the tests: `ranks()` / `nodes()` walk the tree in the construction (pre-)order
// and report (start, size, rank) per internal node.  Node keeps those private and hodlr.h must stay
// untouched, so the header is included with `private` opened up, after every standard header it
// pulls in has been included normally.
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>

#include <cmath>
#include <random>
#include <stdexcept>
#include <vector>
#include <Eigen/Dense>

#include "george/kernels.h"
#include "george/parser.h"
#include "george/exceptions.h"

#define private public
#include "george/hodlr.h"
#undef private

namespace py = pybind11;

class SolverMatrix {                                    // _hodlr.cpp:13-36
public:
  SolverMatrix(george::kernels::Kernel* kernel) : kernel_(kernel), n_(0), ndim_(0) {}
  void set_input_coordinates(const double* x, size_t n, size_t ndim) {
    if (ndim != kernel_->get_ndim()) throw george::dimension_mismatch();
    t_.assign(x, x + n * ndim);
    n_ = n;
    ndim_ = ndim;
  }
  double get_value(const int i, const int j) {
    if (i < 0 || size_t(i) >= n_ || j < 0 || size_t(j) >= n_)
      throw std::out_of_range("attempting to index outside of the dimension of the input coordinates");
    return kernel_->value(&t_[size_t(i) * ndim_], &t_[size_t(j) * ndim_]);
  }
private:
  george::kernels::Kernel* kernel_;
  std::vector<double> t_;
  size_t n_, ndim_;
};

typedef george::hodlr::Node<SolverMatrix> RefNode;

class Solver {
public:
  Solver() : log_det_(0.0), size_(0), computed_(0), kernel_(NULL), matrix_(NULL), solver_(NULL) {}
  ~Solver() {
    if (solver_ != NULL) delete solver_;
    if (matrix_ != NULL) delete matrix_;
    if (kernel_ != NULL) delete kernel_;
  }
  int get_computed() const { return computed_; }
  double log_determinant() const { return log_det_; }
  int size() const { return size_; }

  int compute(const py::object& kernel_spec, py::array_t<double, py::array::c_style | py::array::forcecast> x,
              py::array_t<double, py::array::c_style | py::array::forcecast> yerr, int min_size, double tol, int seed) {
    computed_ = 0;
    if (solver_ != NULL) { delete solver_; solver_ = NULL; }
    if (matrix_ != NULL) { delete matrix_; matrix_ = NULL; }
    if (kernel_ != NULL) { delete kernel_; kernel_ = NULL; }
    kernel_ = george::parse_kernel_spec(kernel_spec);
    matrix_ = new SolverMatrix(kernel_);

    std::mt19937 random;                                // _hodlr.cpp:66-68
    random.seed(seed);

    if (x.ndim() != 2 || yerr.ndim() != 1) throw std::invalid_argument("x must be (n, ndim), yerr (n,)");
    size_t n = size_t(x.shape(0)), ndim = size_t(x.shape(1));
    diag_ = Eigen::VectorXd(Eigen::Index(n));
    const double* ye = yerr.data();
    for (size_t i = 0; i < n; ++i) diag_(Eigen::Index(i)) = ye[i] * ye[i];      // _hodlr.cpp:76
    matrix_->set_input_coordinates(x.data(), n, ndim);

    solver_ = new RefNode(diag_, matrix_, 0, int(n), min_size, tol, random);     // _hodlr.cpp:84-85
    solver_->compute();
    log_det_ = solver_->log_determinant();
    computed_ = 1;
    size_ = int(n);
    return 0;
  }

  // K^-1 applied to an (n,) or (n, nrhs) array; like the reference binding (Eigen::MatrixXd by value)
  // the result is always 2-D
  py::array_t<double> apply_inverse(py::array_t<double, py::array::c_style | py::array::forcecast> y, bool) {
    if (!computed_) throw george::not_computed();
    if (y.ndim() < 1 || y.ndim() > 2 || y.shape(0) != size_) throw george::dimension_mismatch();
    Eigen::Index n = size_, nrhs = y.ndim() == 2 ? y.shape(1) : 1;
    Eigen::MatrixXd b(n, nrhs);
    const double* p = y.data();
    for (Eigen::Index i = 0; i < n; ++i)
      for (Eigen::Index j = 0; j < nrhs; ++j) b(i, j) = p[i * nrhs + j];
    solver_->solve(b);
    py::array_t<double> out({size_t(n), size_t(nrhs)});
    double* o = out.mutable_data();
    for (Eigen::Index i = 0; i < n; ++i)
      for (Eigen::Index j = 0; j < nrhs; ++j) o[i * nrhs + j] = b(i, j);
    return out;
  }

  double dot_solve(py::array_t<double, py::array::c_style | py::array::forcecast> y) {
    if (!computed_) throw george::not_computed();
    if (y.ndim() != 1 || y.shape(0) != size_) throw george::dimension_mismatch();
    Eigen::MatrixXd b(size_, 1);
    for (int i = 0; i < size_; ++i) b(i, 0) = y.data()[i];
    solver_->solve(b);
    double s = 0.0;
    for (int i = 0; i < size_; ++i) s += y.data()[i] * b(i, 0);
    return s;
  }

  py::array_t<double> get_inverse() {
    if (!computed_) throw george::not_computed();
    Eigen::MatrixXd eye(size_, size_);
    eye.setIdentity();
    solver_->solve(eye);
    py::array_t<double> out({size_t(size_), size_t(size_)});
    double* o = out.mutable_data();
    for (int i = 0; i < size_; ++i)
      for (int j = 0; j < size_; ++j) o[size_t(i) * size_ + j] = eye(i, j);
    return out;
  }

  // (level, start, size, rank) of every internal node, in construction (pre-)order
  std::vector<std::vector<int> > nodes() const {
    std::vector<std::vector<int> > out;
    if (solver_ != NULL) walk(solver_, 0, out);
    return out;
  }

private:
  static void walk(const RefNode* nd, int level, std::vector<std::vector<int> >& out) {
    if (nd->is_leaf_) return;
    out.push_back({level, nd->start_, nd->size_, nd->rank_});
    walk(nd->children_[0], level + 1, out);
    walk(nd->children_[1], level + 1, out);
  }

  double log_det_;
  int size_;
  int computed_;
  Eigen::VectorXd diag_;                                // Node keeps a reference to it (hodlr.h:16)
  george::kernels::Kernel* kernel_;
  SolverMatrix* matrix_;
  RefNode* solver_;
};

// Test hooks for oracle/mini_eigen itself (tests/test_oracle_hodlr.py checks these two routines against
// SciPy / NumPy on random matrices: a pivot-order slip in the stand-in would move every HODLR golden).
static Eigen::MatrixXd to_eigen(py::array_t<double, py::array::c_style | py::array::forcecast> a) {
  if (a.ndim() != 2) throw std::invalid_argument("2-D array expected");
  Eigen::MatrixXd m(a.shape(0), a.shape(1));
  for (Eigen::Index i = 0; i < m.rows(); ++i)
    for (Eigen::Index j = 0; j < m.cols(); ++j) m(i, j) = a.data()[i * m.cols() + j];
  return m;
}
static py::array_t<double> from_eigen(const Eigen::MatrixXd& m) {
  py::array_t<double> out({size_t(m.rows()), size_t(m.cols())});
  for (Eigen::Index i = 0; i < m.rows(); ++i)
    for (Eigen::Index j = 0; j < m.cols(); ++j) out.mutable_data()[i * m.cols() + j] = m(i, j);
  return out;
}
static py::tuple mini_eigen_ldlt(py::array_t<double, py::array::c_style | py::array::forcecast> a,
                                 py::array_t<double, py::array::c_style | py::array::forcecast> b) {
  Eigen::LDLT<Eigen::MatrixXd> f;                       // as hodlr.h:24,227,242 uses it
  f.compute(to_eigen(a));
  Eigen::VectorXd d = f.vectorD();
  std::vector<double> dv(size_t(d.rows()));
  for (Eigen::Index i = 0; i < d.rows(); ++i) dv[size_t(i)] = d(i);
  return py::make_tuple(dv, from_eigen(f.solve(to_eigen(b))));
}
static py::tuple mini_eigen_fullpivlu(py::array_t<double, py::array::c_style | py::array::forcecast> a,
                                      py::array_t<double, py::array::c_style | py::array::forcecast> b) {
  Eigen::FullPivLU<Eigen::MatrixXd> f;                  // as hodlr.h:23,233,250 uses it
  f.compute(to_eigen(a));
  return py::make_tuple(from_eigen(f.matrixLU()), int(f.rank()), from_eigen(f.solve(to_eigen(b))));
}

PYBIND11_MODULE(_hodlr, m) {
  m.def("_mini_eigen_ldlt", &mini_eigen_ldlt);
  m.def("_mini_eigen_fullpivlu", &mini_eigen_fullpivlu);
  py::class_<Solver> solver(m, "HODLRSolver");
  solver.def(py::init());
  solver.def_property_readonly("computed", &Solver::get_computed);
  solver.def_property_readonly("log_determinant", &Solver::log_determinant);
  solver.def("compute", &Solver::compute, py::arg("kernel_spec"), py::arg("x"), py::arg("yerr"),
             py::arg("min_size") = 100, py::arg("tol") = 0.1, py::arg("seed") = 42);
  solver.def("apply_inverse", &Solver::apply_inverse, py::arg("x"), py::arg("in_place") = false);
  solver.def("dot_solve", &Solver::dot_solve);
  solver.def("get_inverse", &Solver::get_inverse);
  solver.def("nodes", &Solver::nodes);
}
