// Bandwidth-bound helpers around the conv stack: conv1_1 im2col, 2x2 max-pool, and the
// fp32 <-> halo-tensor conversions at the boundary.  16 bytes per lane everywhere.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "conv_mfma.h"

namespace ron {
namespace {

template <class T> struct Cvt;
template <> struct Cvt<__hip_bfloat16> {
  static __device__ __forceinline__ float to_f(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 from_f(float v) { return __float2bfloat16(v); }
};
template <> struct Cvt<_Float16> {
  static __device__ __forceinline__ float to_f(_Float16 v) { return (float)v; }
  static __device__ __forceinline__ _Float16 from_f(float v) { return (_Float16)v; }
};
template <> struct Cvt<float> {
  static __device__ __forceinline__ float to_f(float v) { return v; }
  static __device__ __forceinline__ float from_f(float v) { return v; }
};

template <class T, int V> struct alignas(sizeof(T) * V) Vec { T v[V]; };

// How a kernel of this file reads / writes V adjacent channels of a tensor whose elements are T; offsets count ELEMENTS from the
// tensor's base (multiples of V for the vector forms).  One 16-byte access per plane.
template <class T> struct IO {
  static constexpr int V = 16 / sizeof(T);
  static constexpr int kEsz = sizeof(T);
  static __device__ __forceinline__ void mode_on() {}          // see IO<SplitF16>
  static __device__ __forceinline__ void load(const void* base, long long off, float* f) {
    const Vec<T, V> a = *reinterpret_cast<const Vec<T, V>*>(reinterpret_cast<const T*>(base) + off);
#pragma unroll
    for (int e = 0; e < V; ++e) f[e] = Cvt<T>::to_f(a.v[e]);
  }
  static __device__ __forceinline__ void store(void* base, long long off, const float* f) {
    Vec<T, V> o;
#pragma unroll
    for (int e = 0; e < V; ++e) o.v[e] = Cvt<T>::from_f(f[e]);
    *reinterpret_cast<Vec<T, V>*>(reinterpret_cast<T*>(base) + off) = o;
  }
  static __device__ __forceinline__ float load1(const void* base, long long off) { return Cvt<T>::to_f(reinterpret_cast<const T*>(base)[off]); }
  static __device__ __forceinline__ void store1(void* base, long long off, float f) { reinterpret_cast<T*>(base)[off] = Cvt<T>::from_f(f); }
};
// RON_DTYPE_F16X3 (conv_device.h, TraitsF16X3S): 32 elements = [32 x f16 hi][32 x f16 lo], value = hi + lo
struct SplitF16 {};
template <> struct IO<SplitF16> {
  static constexpr int V = 8;
  static constexpr int kEsz = 4;
  static __device__ __forceinline__ long long hidx(long long off) { return ((off >> 5) << 6) + (off & 31); }
  // kernels that store this format set MODE.FP16_OVFL first (mode_on): an overflowing conversion saturates at 65504 instead of
  // producing inf - inf = NaN on the way back (conv_device.h, split_mode_on)
  static __device__ __forceinline__ void mode_on() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }
  static __device__ __forceinline__ void split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
  }
  static __device__ __forceinline__ void load(const void* base, long long off, float* f) {
    const _Float16* h = reinterpret_cast<const _Float16*>(base) + hidx(off);
    const Vec<_Float16, 8> hi = *reinterpret_cast<const Vec<_Float16, 8>*>(h), lo = *reinterpret_cast<const Vec<_Float16, 8>*>(h + 32);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)hi.v[e] + (float)lo.v[e];
  }
  static __device__ __forceinline__ void store(void* base, long long off, const float* f) {
    Vec<_Float16, 8> hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      split(f[e], hi.v[e], lo.v[e]);
    }
    _Float16* h = reinterpret_cast<_Float16*>(base) + hidx(off);
    *reinterpret_cast<Vec<_Float16, 8>*>(h) = hi;
    *reinterpret_cast<Vec<_Float16, 8>*>(h + 32) = lo;
  }
  static __device__ __forceinline__ float load1(const void* base, long long off) {
    const _Float16* h = reinterpret_cast<const _Float16*>(base) + hidx(off);
    return (float)h[0] + (float)h[32];
  }
  static __device__ __forceinline__ void store1(void* base, long long off, float f) {
    _Float16* h = reinterpret_cast<_Float16*>(base) + hidx(off);
    _Float16 hi, lo;
    split(f, hi, lo);
    h[0] = hi;
    h[32] = lo;
  }
};

// conv1_1 (Cin = 3): x fp32 [N,H,W,3] -> out T [N,H,W,kc], out[.., (ky*3+kx)*3 + c] = x[y+ky-1, x+kx-1, c]
// (zero outside the image, zero for k >= 27).  The 3x3 conv then runs as a 1x1 conv on the MFMA path.
template <class T>
__global__ void im2col_c3_kernel(const float* __restrict__ x, int n, int h, int w, int kc, void* __restrict__ out) {
  IO<T>::mode_on();
  constexpr int V = IO<T>::V;
  const int groups = kc / V;
  const long long total = (long long)n * h * w * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long long pix = i / groups;
    const int px = (int)(pix % w);
    const int py = (int)((pix / w) % h);
    const long long img = pix / ((long long)w * h);
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int k = g * V + e;
      float val = 0.f;
      if (k < 27) {
        const int tap = k / 3, c = k - tap * 3;
        const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) val = x[((img * h + yy) * w + xx) * 3 + c];
      }
      o[e] = val;
    }
    IO<T>::store(out, i * V, o);
  }
}

struct ViewDev {
  char* base;
  int N, H, W, C, pad, cstride, coff, Hp, Wp;
};
ViewDev to_dev(const TensorView& v) {
  return ViewDev{(char*)v.base, v.N, v.H, v.W, v.C, v.pad, v.cstride, v.coff, v.Hp(), v.Wp()};
}
__device__ __forceinline__ long long view_off(const ViewDev& v, long long img, int y, int x) {
  return ((img * v.Hp + y + v.pad) * v.Wp + x + v.pad) * v.cstride + v.coff;
}

// slim.max_pool2d [2,2] stride 2 (nets/ron_vgg_320.py:456..475); all RON maps are even so SAME == VALID.
template <class T>
__global__ void maxpool2x2_kernel(ViewDev in, ViewDev out) {
  IO<T>::mode_on();
  constexpr int V = IO<T>::V;
  const int groups = out.C / V;
  const long long total = (long long)out.N * out.H * out.W * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long long pix = i / groups;
    const int ox = (int)(pix % out.W);
    const int oy = (int)((pix / out.W) % out.H);
    const long long img = pix / ((long long)out.W * out.H);
    const long long o00 = view_off(in, img, 2 * oy, 2 * ox) + g * V;
    const long long rowstep = (long long)in.Wp * in.cstride;
    float a[V], b[V], c[V], d[V], o[V];
    IO<T>::load(in.base, o00, a);
    IO<T>::load(in.base, o00 + in.cstride, b);
    IO<T>::load(in.base, o00 + rowstep, c);
    IO<T>::load(in.base, o00 + rowstep + in.cstride, d);
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    IO<T>::store(out.base, view_off(out, img, oy, ox) + g * V, o);
  }
}

// slim.max_pool2d [3,3] stride 1 SAME (nets/ssd_vgg_512.py:391, pool5 of SSD).  Inputs are post-ReLU (>= 0), so the
// zero halo is equivalent to TF's "ignore the padding" for a max.
template <class T>
__global__ void maxpool3x3s1_kernel(ViewDev in, ViewDev out) {
  IO<T>::mode_on();
  constexpr int V = IO<T>::V;
  const int groups = out.C / V;
  const long long total = (long long)out.N * out.H * out.W * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const long long pix = i / groups;
    const int ox = (int)(pix % out.W);
    const int oy = (int)((pix / out.W) % out.H);
    const long long img = pix / ((long long)out.W * out.H);
    float m[V];
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        float a[V];
        IO<T>::load(in.base, view_off(in, img, oy + dy, ox + dx) + g * V, a);
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], a[e]);
      }
    IO<T>::store(out.base, view_off(out, img, oy, ox) + g * V, m);
  }
}

// custom_layers.l2_normalization(scaling=True) over the channel axis (nets/custom_layers.py:66-135):
// y = x * rsqrt(max(sum_c x^2, 1e-12)) * gamma[c].  One wave per pixel, 16 B per lane (and plane) per pass.
template <class T>
__global__ void l2norm_kernel(ViewDev in, ViewDev out, const float* __restrict__ gamma) {
  IO<T>::mode_on();
  constexpr int V = IO<T>::V;
  const int lane = threadIdx.x & 63;
  const long long n_pix = (long long)in.N * in.H * in.W;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long pix = wave0; pix < n_pix; pix += n_waves) {
    const int px = (int)(pix % in.W);
    const int py = (int)((pix / in.W) % in.H);
    const long long img = pix / ((long long)in.W * in.H);
    const long long ioff = view_off(in, img, py, px), ooff = view_off(out, img, py, px);
    float ss = 0.f;
    for (int c0 = lane * V; c0 < in.C; c0 += 64 * V) {
      float a[V];
      IO<T>::load(in.base, ioff + c0, a);
#pragma unroll
      for (int e = 0; e < V; ++e) ss += a[e] * a[e];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d, 64);
    const float inv = 1.f / sqrtf(fmaxf(ss, 1e-12f));
    for (int c0 = lane * V; c0 < in.C; c0 += 64 * V) {
      float a[V];
      IO<T>::load(in.base, ioff + c0, a);
#pragma unroll
      for (int e = 0; e < V; ++e) a[e] = a[e] * inv * gamma[c0 + e];
      IO<T>::store(out.base, ooff + c0, a);
    }
  }
}

template <class T>
__global__ void pack_kernel(const float* __restrict__ x, ViewDev out) {
  IO<T>::mode_on();
  const long long total = (long long)out.N * out.H * out.W * out.C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % out.C);
    const long long pix = i / out.C;
    const int px = (int)(pix % out.W);
    const int py = (int)((pix / out.W) % out.H);
    const long long img = pix / ((long long)out.W * out.H);
    IO<T>::store1(out.base, view_off(out, img, py, px) + c, x[i]);
  }
}

template <class T>
__global__ void unpack_kernel(ViewDev in, float* __restrict__ y) {
  const long long total = (long long)in.N * in.H * in.W * in.C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % in.C);
    const long long pix = i / in.C;
    const int px = (int)(pix % in.W);
    const int py = (int)((pix / in.W) % in.H);
    const long long img = pix / ((long long)in.W * in.H);
    y[i] = IO<T>::load1(in.base, view_off(in, img, py, px) + c);
  }
}

template <class T>
__global__ void fill_random_kernel(ViewDev out, unsigned seed) {
  IO<T>::mode_on();
  const long long total = (long long)out.N * out.H * out.W * out.C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % out.C);
    const long long pix = i / out.C;
    const int px = (int)(pix % out.W);
    const int py = (int)((pix / out.W) % out.H);
    const long long img = pix / ((long long)out.W * out.H);
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // uniform [-1, 1); the split-precision form gets low-order bits too so that its lo plane is not all zeros
    const float v = IO<T>::kEsz == 4 ? (float)(h & 0xFFFFFF) / 8388608.f - 1.f : (float)(h & 0xFFFF) / 32768.f - 1.f;
    IO<T>::store1(out.base, view_off(out, img, py, px) + c, v);
  }
}

int grid_for(long long total) { return (int)std::min<long long>((total + 255) / 256, 256 * 8); }
int vec_elems(int dtype) { return dtype == RON_DTYPE_F16X3 ? 8 : 16 / (int)dtype_size(dtype); }   // IO<T>::V of the dtype

// launches KERNEL<T> for the element type of `dtype`
#define RON_DISPATCH_DTYPE(dtype, KERNEL, grid, stream, ...)                                                               \
  do {                                                                                                                     \
    if ((dtype) == RON_DTYPE_BF16) RON_LAUNCH(KERNEL<__hip_bfloat16>, dim3(grid), dim3(256), 0, stream, __VA_ARGS__); \
    else if ((dtype) == RON_DTYPE_F16) RON_LAUNCH(KERNEL<_Float16>, dim3(grid), dim3(256), 0, stream, __VA_ARGS__);  \
    else if ((dtype) == RON_DTYPE_F16X3) RON_LAUNCH(KERNEL<SplitF16>, dim3(grid), dim3(256), 0, stream, __VA_ARGS__); \
    else RON_LAUNCH(KERNEL<float>, dim3(grid), dim3(256), 0, stream, __VA_ARGS__);                                  \
  } while (0)

}  // namespace

int launch_im2col_c3(const float* x, int n, int h, int w, int dtype, void* out, int kchunk, hipStream_t s) {
  const long long total = (long long)n * h * w * (kchunk / vec_elems(dtype));
  const int g = grid_for(total);
  RON_DISPATCH_DTYPE(dtype, im2col_c3_kernel, g, s, x, n, h, w, kchunk, out);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

int launch_maxpool2x2(const TensorView& in, const TensorView& out, int dtype, hipStream_t s) {
  const int V = vec_elems(dtype);
  RON_REQUIRE(in.H == 2 * out.H && in.W == 2 * out.W && in.C == out.C && in.N == out.N, "maxpool: shape mismatch");
  RON_REQUIRE(out.C % V == 0 && in.cstride % V == 0 && out.cstride % V == 0 && in.coff % V == 0 && out.coff % V == 0,
              "maxpool: channels must be a multiple of %d", V);
  const long long total = (long long)out.N * out.H * out.W * (out.C / V);
  const int g = grid_for(total);
  RON_DISPATCH_DTYPE(dtype, maxpool2x2_kernel, g, s, to_dev(in), to_dev(out));
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

int launch_maxpool3x3s1(const TensorView& in, const TensorView& out, int dtype, hipStream_t s) {
  const int V = vec_elems(dtype);
  RON_REQUIRE(in.H == out.H && in.W == out.W && in.C == out.C && in.N == out.N && in.pad >= 1, "maxpool3x3: shape / halo mismatch");
  RON_REQUIRE(out.C % V == 0 && in.cstride % V == 0 && out.cstride % V == 0, "maxpool3x3: channels must be a multiple of %d", V);
  const int g = grid_for((long long)out.N * out.H * out.W * (out.C / V));
  RON_DISPATCH_DTYPE(dtype, maxpool3x3s1_kernel, g, s, to_dev(in), to_dev(out));
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

int launch_l2norm(const TensorView& in, const TensorView& out, const float* d_gamma, int dtype, hipStream_t s) {
  const int V = vec_elems(dtype);
  RON_REQUIRE(in.H == out.H && in.W == out.W && in.C == out.C && in.N == out.N && in.C % V == 0, "l2norm: shape mismatch");
  const long long waves = (long long)in.N * in.H * in.W;
  const int g = (int)std::min<long long>((waves + 3) / 4, 256 * 8);
  RON_DISPATCH_DTYPE(dtype, l2norm_kernel, g, s, to_dev(in), to_dev(out), d_gamma);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

// split precision: the element index -> plane index mapping wants 32-element alignment of every pixel / channel slice
static int check_split_view(const TensorView& v, int dtype) {
  if (dtype == RON_DTYPE_F16X3) RON_REQUIRE(v.cstride % 32 == 0 && v.coff % 32 == 0, "split-precision tensors: pixel stride and channel slices must be multiples of 32");
  return RON_OK;
}

int launch_pack_input(const float* x, const TensorView& out, int dtype, hipStream_t s) {
  int rc = check_split_view(out, dtype);
  if (rc) return rc;
  const int g = grid_for((long long)out.N * out.H * out.W * out.C);
  RON_DISPATCH_DTYPE(dtype, pack_kernel, g, s, x, to_dev(out));
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

int launch_fill_random(const TensorView& out, int dtype, unsigned seed, hipStream_t s) {
  int rc = check_split_view(out, dtype);
  if (rc) return rc;
  const int g = grid_for((long long)out.N * out.H * out.W * out.C);
  RON_DISPATCH_DTYPE(dtype, fill_random_kernel, g, s, to_dev(out), seed);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

int launch_unpack(const TensorView& in, int dtype, int in_is_f32, float* y, hipStream_t s) {
  int rc = check_split_view(in, in_is_f32 ? RON_DTYPE_F32 : dtype);
  if (rc) return rc;
  const int g = grid_for((long long)in.N * in.H * in.W * in.C);
  RON_DISPATCH_DTYPE(in_is_f32 ? RON_DTYPE_F32 : dtype, unpack_kernel, g, s, to_dev(in), y);
  RON_HIP_CHECK(ron::launch_error());
  return RON_OK;
}

}  // namespace ron
