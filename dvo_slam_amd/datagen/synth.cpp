/*
 * synth.cpp -- deterministic synthetic RGB-D frame pairs.
 *
 * A data tool, not part of the alignment path and not part of the oracle: there is no TUM dataset on
 * disk and no network, so the parity tests, the bench workload and the CPU baseline all consume these
 * frames (CPU and GPU sides read identical bytes).  Built as libdvo_synth.so by plain g++.  The generator
 * renders the SAME analytic scene from two camera poses (exact re-render by ray/surface root
 * finding, not image warping), then quantises exactly like the reference's ingest does:
 * grey -> u8 (stored as float 0..255 by the caller), depth -> u16 at 5000 counts per metre, 0 = hole
 * (dvo_benchmark/src/benchmark_slam.cpp:46-93, dvo_core/src/core/surface_pyramid.cpp:65-105).
 */

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <thread>
#include <vector>

namespace {

struct Pcg32 {
  uint64_t state, inc;
  explicit Pcg32(uint64_t seed, uint64_t seq = 54u) {
    state = 0u;
    inc = (seq << 1u) | 1u;
    next();
    state += seed;
    next();
  }
  uint32_t next() {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = uint32_t(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
  }
  double uniform() { return (next() >> 8) * (1.0 / 16777216.0); }   // [0,1)
  double uniform(double a, double b) { return a + (b - a) * uniform(); }
  double gaussian() {                                                 // Box-Muller, one value per call (the twin is dropped)
    const double u1 = 1.0 - uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
};

struct Scene {
  int W, H;
  double fx, fy, ox, oy;
  double dir[6][3], k[6], phase[6], amp[6];
  double tex_lo, tex_hi;

  double depth_at(double u, double v) const {   // analytic reference-view depth, defined for any real (u,v)
    return 1.5 + 0.5 * std::sin(2.0 * M_PI * u / W * 1.5) * std::cos(2.0 * M_PI * v / H) + 0.3 * (v / H);
  }
  double texture_raw(const double p[3]) const {
    double s = 0;
    for (int i = 0; i < 6; ++i) s += amp[i] * std::sin(k[i] * (dir[i][0] * p[0] + dir[i][1] * p[1] + dir[i][2] * p[2]) + phase[i]);
    return s;
  }
  double texture(const double p[3]) const {   // mapped to 20..235 grey levels
    double t = (texture_raw(p) - tex_lo) / (tex_hi - tex_lo);
    return 20.0 + 215.0 * t;
  }
};

void make_scene(Scene& sc, Pcg32& rng) {
  static const double wavelengths[6] = {0.05, 0.09, 0.15, 0.25, 0.4, 0.6};
  double asum = 0;
  for (int i = 0; i < 6; ++i) {
    double d[3] = {rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-0.3, 0.3)};
    double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-9;
    for (int j = 0; j < 3; ++j) sc.dir[i][j] = d[j] / n;
    sc.k[i] = 2.0 * M_PI / wavelengths[i];
    sc.phase[i] = rng.uniform(0, 2.0 * M_PI);
    sc.amp[i] = 0.5 + 0.5 * double(i) / 5.0;   // longer wavelengths carry more contrast
    asum += sc.amp[i];
  }
  sc.tex_lo = -0.75 * asum;   // the sum rarely reaches +-asum; clip the tails instead of wasting range
  sc.tex_hi = 0.75 * asum;
}

inline uint8_t quantise_grey(double g) {
  if (g < 0) g = 0;
  if (g > 255) g = 255;
  return uint8_t(std::lround(g));
}

inline uint16_t quantise_depth(double z) {
  double q = std::round(z * 5000.0);
  if (!(q >= 1.0)) return 0;
  if (q > 65535.0) return 0;
  return uint16_t(q);
}

// exp of the twist (upsilon, omega) as a row-major 4x4 (Rodrigues; |omega| here is 0.009..0.03 rad)
void se3_exp_matrix(const double x[6], double M[16]) {
  const double* u = x;
  const double* w = x + 3;
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
  double a, b, c;
  if (th < 1e-6) { a = 1.0 - th2 / 6.0; b = 0.5 - th2 / 24.0; c = 1.0 / 6.0 - th2 / 120.0; }
  else { a = std::sin(th) / th; b = (1.0 - std::cos(th)) / th2; c = (th - std::sin(th)) / (th2 * th); }
  const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
  for (int i = 0; i < 3; ++i) {
    double t = 0;
    for (int j = 0; j < 3; ++j) {
      const double I = i == j ? 1.0 : 0.0;
      M[i * 4 + j] = I + a * O[i * 3 + j] + b * O2[i * 3 + j];
      t += (I + b * O[i * 3 + j] + c * O2[i * 3 + j]) * u[j];
    }
    M[i * 4 + 3] = t;
  }
  M[12] = M[13] = M[14] = 0;
  M[15] = 1;
}

void punch_holes(uint16_t* depth, int W, int H, Pcg32& rng) {
  for (int by = 0; by < H; by += 8)
    for (int bx = 0; bx < W; bx += 8)
      if (rng.uniform() < 0.12)
        for (int y = by; y < by + 8 && y < H; ++y)
          for (int x = bx; x < bx + 8 && x < W; ++x) depth[size_t(y) * W + x] = 0;
  const int band = (10 * W) / 640 > 0 ? (10 * W) / 640 : 1;   // Kinect-style shadow band on the right
  for (int y = 0; y < H; ++y)
    for (int x = W - band; x < W; ++x) depth[size_t(y) * W + x] = 0;
}

}  // namespace

extern "C" void dvo_synth_pair(uint64_t seed, int W, int H, const float K[4], uint8_t* grey_ref, uint16_t* depth_ref,
                                  uint8_t* grey_cur, uint16_t* depth_cur, double xi_true[6]) {
  Pcg32 rng(seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL, seed + 7);
  Scene sc;
  sc.W = W; sc.H = H;
  sc.fx = K[0]; sc.fy = K[1]; sc.ox = K[2]; sc.oy = K[3];
  make_scene(sc, rng);

  // motion: |v| <= 0.03 m, |omega| <= 0.03 rad (typical 30 Hz fr1 inter-frame motion)
  double xi[6];
  for (int part = 0; part < 2; ++part) {
    double d[3] = {rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)};
    double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + 1e-9;
    double mag = 0.03 * rng.uniform(0.3, 1.0);
    for (int j = 0; j < 3; ++j) xi[part * 3 + j] = d[j] / n * mag;
  }
  for (int i = 0; i < 6; ++i) xi_true[i] = xi[i];
  double Mrc[16];   // current -> reference: the transform match() should return
  se3_exp_matrix(xi, Mrc);

  Pcg32 noise_ref(seed * 31 + 1, 11), noise_cur(seed * 31 + 2, 13);
  Pcg32 holes_ref(seed * 131 + 5, 17), holes_cur(seed * 131 + 6, 19);

  // reference view
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const double z = sc.depth_at(u, v);
      const double p[3] = {(u - sc.ox) / sc.fx * z, (v - sc.oy) / sc.fy * z, z};
      grey_ref[size_t(v) * W + u] = quantise_grey(sc.texture(p) + noise_ref.uniform(-1.5, 1.5));
      depth_ref[size_t(v) * W + u] = quantise_depth(z);
    }
  punch_holes(depth_ref, W, H, holes_ref);

  // current view: intersect each pixel ray with the surface  p.z = depth_at(project(p)),  p = R s d + t
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const double d[3] = {(u - sc.ox) / sc.fx, (v - sc.oy) / sc.fy, 1.0};
      double rd[3], t[3] = {Mrc[3], Mrc[7], Mrc[11]};
      for (int i = 0; i < 3; ++i) rd[i] = Mrc[i * 4 + 0] * d[0] + Mrc[i * 4 + 1] * d[1] + Mrc[i * 4 + 2] * d[2];
      auto g = [&](double s, double p[3]) {
        for (int i = 0; i < 3; ++i) p[i] = rd[i] * s + t[i];
        const double pu = sc.fx * p[0] / p[2] + sc.ox, pv = sc.fy * p[1] / p[2] + sc.oy;
        return p[2] - sc.depth_at(pu, pv);
      };
      double s = sc.depth_at(u, v), p[3];
      for (int it = 0; it < 20; ++it) {
        const double g0 = g(s, p);
        if (std::fabs(g0) < 1e-9) break;
        double ph[3];
        const double h = 1e-4;
        const double g1 = g(s + h, ph);
        double dg = (g1 - g0) / h;
        if (std::fabs(dg) < 0.05) dg = dg < 0 ? -0.05 : 0.05;
        double step = g0 / dg;
        if (step > 0.2) step = 0.2;
        if (step < -0.2) step = -0.2;
        s -= step;
      }
      g(s, p);
      grey_cur[size_t(v) * W + u] = quantise_grey(sc.texture(p) + noise_cur.uniform(-1.5, 1.5));
      depth_cur[size_t(v) * W + u] = quantise_depth(s);
    }
  punch_holes(depth_cur, W, H, holes_cur);
}

// n pairs with seeds seed0 .. seed0+n-1 into contiguous arrays, generated on `nthreads` host threads
extern "C" void dvo_synth_batch(uint64_t seed0, int n, int W, int H, const float K[4], uint8_t* grey_ref, uint16_t* depth_ref,
                                uint8_t* grey_cur, uint16_t* depth_cur, double* xi_true, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  const size_t npx = size_t(W) * H;
  auto work = [&](int t) {
    for (int i = t; i < n; i += nthreads)
      dvo_synth_pair(seed0 + uint64_t(i), W, H, K, grey_ref + i * npx, depth_ref + i * npx, grey_cur + i * npx, depth_cur + i * npx,
                     xi_true + size_t(i) * 6);
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

namespace {

// one view of the scene from the camera whose camera->world transform is Mcw (world = the frame the analytic surface
// is defined in): intersect each pixel ray with the surface  p.z = depth_at(project(p)),  p = R s d + t
// depth_noise: multiples of the Kinect depth uncertainty sigma(z) = 0.0012 + 0.0019 (z - 0.4)^2 m -- the sensor model the
// reference itself assumes (dvo_core/src/dense_tracking_impl.cpp:122-128) -- added as zero-mean Gaussian noise before the
// 1/5000 m quantisation; grey_noise: half-width of the uniform intensity noise in grey levels.
void render_view(const Scene& sc, const double Mcw[16], Pcg32& noise, Pcg32& holes, uint8_t* grey, uint16_t* depth,
                 double depth_noise = 0.0, double grey_noise = 1.5, double gain = 1.0, double bias = 0.0) {
  const int W = sc.W, H = sc.H;
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const double d[3] = {(u - sc.ox) / sc.fx, (v - sc.oy) / sc.fy, 1.0};
      double rd[3], t[3] = {Mcw[3], Mcw[7], Mcw[11]};
      for (int i = 0; i < 3; ++i) rd[i] = Mcw[i * 4 + 0] * d[0] + Mcw[i * 4 + 1] * d[1] + Mcw[i * 4 + 2] * d[2];
      auto g = [&](double s, double p[3]) {
        for (int i = 0; i < 3; ++i) p[i] = rd[i] * s + t[i];
        const double pu = sc.fx * p[0] / p[2] + sc.ox, pv = sc.fy * p[1] / p[2] + sc.oy;
        return p[2] - sc.depth_at(pu, pv);
      };
      double s = sc.depth_at(u, v), p[3];
      for (int it = 0; it < 30; ++it) {
        const double g0 = g(s, p);
        if (std::fabs(g0) < 1e-9) break;
        double ph[3];
        const double h = 1e-4;
        double dg = (g(s + h, ph) - g0) / h;
        if (std::fabs(dg) < 0.05) dg = dg < 0 ? -0.05 : 0.05;
        double step = g0 / dg;
        if (step > 0.2) step = 0.2;
        if (step < -0.2) step = -0.2;
        s -= step;
      }
      g(s, p);
      grey[size_t(v) * W + u] = quantise_grey(gain * sc.texture(p) + bias + noise.uniform(-grey_noise, grey_noise));
      double z = s;
      if (depth_noise > 0.0) {
        const double sigma = 0.0012 + 0.0019 * (s - 0.4) * (s - 0.4);
        z += depth_noise * sigma * noise.gaussian();
      }
      depth[size_t(v) * W + u] = quantise_depth(z);
    }
  punch_holes(depth, W, H, holes);
}

}  // namespace

// A camera sweep over one scene: n frames, frame k seen from the pose  T_k = exp(xi(k)),  xi_i(k) = A_i sin(w_i k + phi_i)
// (bounded excursion around the scene's defining view, per-frame motion of the order of 30 Hz hand-held footage).
// poses: n row-major 4x4 camera->world transforms = the ground truth trajectory a replay is evaluated against.
// exposure: per-frame auto-exposure drift -- frame k is rendered with gain 1 + exposure * g_k and bias 100 * exposure * b_k grey
// levels, g and b smooth pseudo-random sequences in [-1, 1] (sums of three incommensurate sinusoids): the brightness-constancy
// violation every real sequence has and the photometric model does not know about.
extern "C" void dvo_synth_sequence_noisy(uint64_t seed, int n, int W, int H, const float K[4], uint8_t* grey, uint16_t* depth, double* poses,
                                         double depth_noise, double grey_noise, double exposure, int nthreads) {
  Pcg32 rng(seed * 0x9E3779B97F4A7C15ULL + 0x7654321ULL, seed + 23);
  Scene sc;
  sc.W = W; sc.H = H;
  sc.fx = K[0]; sc.fy = K[1]; sc.ox = K[2]; sc.oy = K[3];
  make_scene(sc, rng);
  double amp[6], freq[6], phase[6];
  for (int i = 0; i < 6; ++i) {
    amp[i] = (i < 3 ? 0.12 : 0.10) * rng.uniform(0.5, 1.0);      // metres / radians
    freq[i] = rng.uniform(0.08, 0.16);                           // rad per frame -> steps of <= ~0.02 per frame
    phase[i] = rng.uniform(0, 2.0 * M_PI);
  }
  for (int k = 0; k < n; ++k) {
    double xi[6];
    for (int i = 0; i < 6; ++i) xi[i] = amp[i] * (std::sin(freq[i] * k + phase[i]) - std::sin(phase[i]));   // frame 0 = identity
    se3_exp_matrix(xi, poses + size_t(k) * 16);
  }
  double ef[6], ep[6];
  for (int i = 0; i < 6; ++i) {
    ef[i] = rng.uniform(0.05, 0.6);                              // rad per frame: exposure changes over 10..120 frames
    ep[i] = rng.uniform(0, 2.0 * M_PI);
  }
  if (nthreads < 1) nthreads = 1;
  const size_t npx = size_t(W) * H;
  auto work = [&](int t) {
    for (int k = t; k < n; k += nthreads) {
      Pcg32 noise(seed * 977 + 2 * uint64_t(k) + 1, 29), holes(seed * 1543 + 2 * uint64_t(k) + 2, 31);
      const double g = (std::sin(ef[0] * k + ep[0]) + std::sin(ef[1] * k + ep[1]) + std::sin(ef[2] * k + ep[2])) / 3.0;
      const double b = (std::sin(ef[3] * k + ep[3]) + std::sin(ef[4] * k + ep[4]) + std::sin(ef[5] * k + ep[5])) / 3.0;
      render_view(sc, poses + size_t(k) * 16, noise, holes, grey + k * npx, depth + k * npx, depth_noise, grey_noise,
                  1.0 + exposure * g, 100.0 * exposure * b);
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}

// the noise-free variant (depth quantisation and +-1.5 grey levels of intensity noise only)
extern "C" void dvo_synth_sequence(uint64_t seed, int n, int W, int H, const float K[4], uint8_t* grey, uint16_t* depth, double* poses,
                                   int nthreads) {
  dvo_synth_sequence_noisy(seed, n, W, H, K, grey, depth, poses, 0.0, 1.5, 0.0, nthreads);
}
