// TEST INFRASTRUCTURE ONLY.  C entry points around the reference's own permutohedral-lattice code
// (/root/reference/wrapper/bilateralfilter/permutohedral.{hpp,cpp}, bilateralfilter.{hpp,cpp}), which oracle/Makefile compiles
// from where it lies into oracle/_ref/libpermuto_ref.so.  Nothing of the reference is copied here: this file only calls it.
// The class keeps its tables protected; a subclass exposes them so the restatement in oracle/crf_oracle.py can be checked
// table by table (offsets, barycentric weights) and not only through filter outputs.
#include "bilateralfilter.hpp"

namespace {
class Probe : public Permutohedral {
 public:
    int n_points() const { return M_; }
    const int* offsets() const { return offset_; }
    const float* weights() const { return barycentric_; }
};
}  // namespace

extern "C" {

// bilateralfilter.cpp:22-41 as is: image (3,H,W) float planes, in / out (K,H,W); one class plane at a time
void ref_bilateralfilter(float* image, float* in, float* out, int K, int H, int W, float sigmargb, float sigmaxy) {
    bilateralfilter(image, 3 * H * W, in, K * H * W, out, K * H * W, H, W, sigmargb, sigmaxy);
}

// Permutohedral::init + compute on caller-made features (N, d) and values (N, value_size), both row-major
int ref_lattice_filter(const float* features, int d, int N, const float* in, float* out, int value_size) {
    Probe lattice;
    lattice.init(features, d, N);
    lattice.compute(out, in, value_size);
    return lattice.n_points();
}

// the tables of init: offsets (N, d+1) = lattice point of every vertex of the enclosing simplex, weights (N, d+1)
int ref_lattice_tables(const float* features, int d, int N, int* offsets, float* weights) {
    Probe lattice;
    lattice.init(features, d, N);
    for (int i = 0; i < N * (d + 1); ++i) {
        offsets[i] = lattice.offsets()[i];
        weights[i] = lattice.weights()[i];
    }
    return lattice.n_points();
}
}
