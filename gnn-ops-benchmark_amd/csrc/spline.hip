// spline.hip — the B-spline convolution ops the reference lists among what it means to benchmark (ops.txt:17-19
// "Spline conv support ops": torch.ops.torch_spline_conv.spline_basis / spline_weighting; ops.txt:29-31
// torch_spline_conv.spline_conv) — SURVEY.md §8(f) rank 4. The package itself (torch-spline-conv 1.2.1,
// requirements.txt:214) is not in the reference tree; the arithmetic below is its published definition (SplineCNN,
// Fey et al. 2018, and the package's basis / weighting kernels): parity unpinned, oracle in oracle/spatial_oracle.py.
//
//   spline_basis(pseudo [E, D] in [0, 1], kernel_size [D], is_open_spline [D], degree m)
//       -> basis [E, S] and weight_index [E, S],  S = (m + 1)^D. For combination s = (k_0, .., k_{D-1}) in base m + 1:
//            v_d = pseudo[e, d] * (kernel_size[d] - m * is_open_spline[d])
//            weight_index += ((floor(v_d) + k_d) mod kernel_size[d]) * prod_{d' < d} kernel_size[d']
//            basis *= B_m(frac(v_d), k_d)            B_1: 1 - v | v;  B_2: v^2/2 - v + 1/2 | -v^2 + v + 1/2 | v^2/2;
//                                                    B_3: (1-v)^3/6 | (3v^3 - 6v^2 + 4)/6 | (-3v^3 + 3v^2 + 3v + 1)/6 | v^3/6
//   spline_weighting(x [E, Min], weight [K, Min, Mout], basis, weight_index) -> out [E, Mout]
//            out[e, o] = sum_s basis[e, s] * sum_i x[e, i] * weight[weight_index[e, s], i, o]
//   spline_conv = the two above on x[col], summed per destination row (and divided by its degree): here ONE pass over the
//   destination-sorted edge list — basis values are recomputed in registers, the [E, S] and [E, Mout] tensors never exist.
//
// A wave owns one (edge | destination row, 64 output channels): the weight rows it reads are 256-B coalesced pieces of
// a table that lives in L2 (K * Min * Mout * 4 B: 125 kernels of 64 x 64 fp32 = 2 MB); x[e, i] and the basis are
// wave-uniform. fp32 accumulation in the order (edge, s, i); VALU FMA work — the contraction is per EDGE with its own blend
// of weight matrices, there is no shared operand tile to put on the matrix cores without first sorting E * S products by
// kernel index.
#include "common.h"

namespace {

constexpr int MAX_DIM = 8;
constexpr int MAX_S = 64;

struct SplineMeta {
    int D, degree, S;
    int64_t kernel_size[MAX_DIM];
    int is_open[MAX_DIM];
};

template <int M>
__device__ inline float bspline(float v, int k) {
    if constexpr (M == 1) {
        return k == 0 ? 1.f - v : v;
    } else if constexpr (M == 2) {
        if (k == 0) return 0.5f * v * v - v + 0.5f;
        if (k == 1) return -v * v + v + 0.5f;
        return 0.5f * v * v;
    } else {
        if (k == 0) return (1.f - v) * (1.f - v) * (1.f - v) / 6.f;
        if (k == 1) return (3.f * v * v * v - 6.f * v * v + 4.f) / 6.f;
        if (k == 2) return (-3.f * v * v * v + 3.f * v * v + 3.f * v + 1.f) / 6.f;
        return v * v * v / 6.f;
    }
}

// basis value and weight index of combination s for one edge's pseudo-coordinates (fp32 arithmetic for every storage type)
template <typename T, int M>
__device__ inline void basis_of(const T* __restrict__ pseudo_e, const SplineMeta& sm, int s, float& b, int64_t& wi) {
    int k = s;
    int64_t off = 1;
    wi = 0;
    b = 1.f;
    for (int d = 0; d < sm.D; ++d) {
        const int k_mod = k % (M + 1);
        k /= (M + 1);
        float v = Elem<T>::load(pseudo_e + d) * (float)(sm.kernel_size[d] - M * sm.is_open[d]);
        const float fl = floorf(v);
        wi += (((int64_t)fl + k_mod) % sm.kernel_size[d]) * off;
        off *= sm.kernel_size[d];
        v -= fl;
        b *= bspline<M>(v, k_mod);
    }
}

template <typename T, int M>
__global__ __launch_bounds__(256) void spline_basis_kernel(const T* __restrict__ pseudo, SplineMeta sm, int64_t E,
                                                           T* __restrict__ basis, int64_t* __restrict__ weight_index) {
    const int64_t total = E * sm.S;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = t / sm.S;
        const int s = (int)(t % sm.S);
        float b;
        int64_t wi;
        basis_of<T, M>(pseudo + e * sm.D, sm, s, b, wi);
        Elem<T>::store(basis + t, b);
        weight_index[t] = wi;
    }
}

// out[e, o] = sum_s basis[e, s] * sum_i x[e, i] * weight[wi[e, s], i, o]: one wave per (e, 64-channel chunk), lane = o
template <typename T>
__global__ __launch_bounds__(256) void spline_weighting_kernel(const T* __restrict__ x, const T* __restrict__ weight,
                                                               const T* __restrict__ basis, const int64_t* __restrict__ weight_index,
                                                               T* __restrict__ out, int64_t E, int Min, int Mout, int S, int ochunks) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t item = wave0; item < E * ochunks; item += nwaves) {
        const int64_t e = item / ochunks;
        const int o = (int)(item % ochunks) * 64 + lane;
        float acc = 0.f;
        for (int s = 0; s < S; ++s) {
            const float b = Elem<T>::load(basis + e * S + s);
            const T* wk = weight + weight_index[e * S + s] * Min * Mout;
            for (int i = 0; i < Min; ++i) {
                const float xv = b * Elem<T>::load(x + e * Min + i);
                if (o < Mout) acc += xv * Elem<T>::load(wk + (int64_t)i * Mout + o);
            }
        }
        if (o < Mout) Elem<T>::store(out + e * Mout + o, acc);
    }
}

// The whole convolution per destination: rowptr / perm = plan of edge_index[0] (the row the messages are summed at),
// src = edge_index[1] in plan order (the row whose features travel), pseudo in ORIGINAL edge order (perm maps back).
template <typename T, int M>
__global__ __launch_bounds__(256) void spline_conv_kernel(const T* __restrict__ x, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ perm, const int64_t* __restrict__ src,
                                                          const T* __restrict__ pseudo, const T* __restrict__ weight, SplineMeta sm,
                                                          const T* __restrict__ root, const T* __restrict__ bias,
                                                          T* __restrict__ out, int64_t N, int Min, int Mout, int ochunks, int norm) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t item = wave0; item < N * ochunks; item += nwaves) {
        const int64_t n = item / ochunks;
        const int o = (int)(item % ochunks) * 64 + lane;
        const int32_t beg = rowptr[n], end = rowptr[n + 1];
        float acc = 0.f;
        for (int32_t j = beg; j < end; ++j) {
            const int64_t col = src[j];
            const int64_t e = perm ? perm[j] : j;
            for (int s = 0; s < sm.S; ++s) {
                float b;
                int64_t wi;
                basis_of<T, M>(pseudo + e * sm.D, sm, s, b, wi);
                const T* wk = weight + wi * Min * Mout;
                for (int i = 0; i < Min; ++i) {
                    const float xv = b * Elem<T>::load(x + col * Min + i);
                    if (o < Mout) acc += xv * Elem<T>::load(wk + (int64_t)i * Mout + o);
                }
            }
        }
        if (norm) {
            const int32_t deg = end - beg;
            acc = acc / (float)(deg < 1 ? 1 : deg);
        }
        if (root) {
            for (int i = 0; i < Min; ++i)
                if (o < Mout) acc += Elem<T>::load(x + n * Min + i) * Elem<T>::load(root + (int64_t)i * Mout + o);
        }
        if (bias && o < Mout) acc += Elem<T>::load(bias + o);
        if (o < Mout) Elem<T>::store(out + n * Mout + o, acc);
    }
}

int fill_meta(SplineMeta& sm, const int64_t* kernel_size, const uint8_t* is_open_spline, int D, int degree, const char* what) {
    GNNOPS_REQUIRE(D >= 1 && D <= MAX_DIM, GNNOPS_EUNSUPPORTED, "%s: 1..%d pseudo-coordinate dimensions", what, MAX_DIM);
    GNNOPS_REQUIRE(degree >= 1 && degree <= 3, GNNOPS_EUNSUPPORTED, "%s: B-spline degree must be 1, 2 or 3", what);
    GNNOPS_REQUIRE(kernel_size && is_open_spline, GNNOPS_EINVAL, "%s: null kernel_size / is_open_spline", what);
    sm.D = D;
    sm.degree = degree;
    int64_t S = 1;
    for (int d = 0; d < D; ++d) {
        S *= degree + 1;
        GNNOPS_REQUIRE(kernel_size[d] >= 1, GNNOPS_EINVAL, "%s: kernel_size must be positive", what);
        sm.kernel_size[d] = kernel_size[d];
        sm.is_open[d] = is_open_spline[d] ? 1 : 0;
    }
    GNNOPS_REQUIRE(S <= MAX_S, GNNOPS_EUNSUPPORTED, "%s: (degree + 1)^D = %lld basis products per edge, at most %d", what, (long long)S,
                   MAX_S);
    sm.S = (int)S;
    return GNNOPS_OK;
}

template <typename T>
int run_basis(const void* pseudo, const SplineMeta& sm, int64_t E, void* basis, int64_t* wi, hipStream_t stream) {
    const int grid = gnnops_grid_cap(gnnops_cdiv(E * sm.S, 256));
    switch (sm.degree) {
        case 1: hipLaunchKernelGGL((spline_basis_kernel<T, 1>), dim3(grid), dim3(256), 0, stream, (const T*)pseudo, sm, E, (T*)basis, wi); break;
        case 2: hipLaunchKernelGGL((spline_basis_kernel<T, 2>), dim3(grid), dim3(256), 0, stream, (const T*)pseudo, sm, E, (T*)basis, wi); break;
        default: hipLaunchKernelGGL((spline_basis_kernel<T, 3>), dim3(grid), dim3(256), 0, stream, (const T*)pseudo, sm, E, (T*)basis, wi); break;
    }
    return gnnops_check_launch("spline_basis");
}

template <typename T>
int run_conv(const void* x, const int32_t* rowptr, const int32_t* perm, const int64_t* src, const void* pseudo, const void* weight,
             const SplineMeta& sm, const void* root, const void* bias, void* out, int64_t N, int Min, int Mout, int norm,
             hipStream_t stream) {
    const int ochunks = (Mout + 63) / 64;
    const int grid = gnnops_grid_cap(gnnops_cdiv(N * ochunks, 4), 256 * 32);
#define GNNOPS_SC(MM)                                                                                                              \
    hipLaunchKernelGGL((spline_conv_kernel<T, MM>), dim3(grid), dim3(256), 0, stream, (const T*)x, rowptr, perm, src, (const T*)pseudo, \
                       (const T*)weight, sm, (const T*)root, (const T*)bias, (T*)out, N, Min, Mout, ochunks, norm)
    switch (sm.degree) {
        case 1: GNNOPS_SC(1); break;
        case 2: GNNOPS_SC(2); break;
        default: GNNOPS_SC(3); break;
    }
#undef GNNOPS_SC
    return gnnops_check_launch("spline_conv");
}

}  // namespace

extern "C" int gnnops_spline_basis(const void* pseudo, const int64_t* kernel_size, const uint8_t* is_open_spline, int64_t E, int D,
                                   int degree, void* basis, int64_t* weight_index, int dtype, gnnops_stream_t s) {
    SplineMeta sm{};
    const int rc = fill_meta(sm, kernel_size, is_open_spline, D, degree, "spline_basis");
    if (rc != GNNOPS_OK) return rc;
    GNNOPS_REQUIRE(E >= 0, GNNOPS_EINVAL, "spline_basis: negative size");
    if (E == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(pseudo && basis && weight_index, GNNOPS_EINVAL, "spline_basis: null pointer");
    switch (dtype) {
        case GNNOPS_F32: return run_basis<float>(pseudo, sm, E, basis, weight_index, (hipStream_t)s);
        case GNNOPS_F16: return run_basis<__half>(pseudo, sm, E, basis, weight_index, (hipStream_t)s);
        case GNNOPS_BF16: return run_basis<__hip_bfloat16>(pseudo, sm, E, basis, weight_index, (hipStream_t)s);
    }
    gnnops_set_error("spline_basis: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}

extern "C" int gnnops_spline_weighting(const void* x, const void* weight, const void* basis, const int64_t* weight_index, void* out,
                                       int64_t E, int64_t Min, int64_t Mout, int64_t S, int dtype, gnnops_stream_t s) {
    GNNOPS_REQUIRE(E >= 0 && Min >= 0 && Mout >= 0 && S >= 0, GNNOPS_EINVAL, "spline_weighting: negative size");
    GNNOPS_REQUIRE(Min < (1 << 20) && Mout < (1 << 20) && S <= MAX_S, GNNOPS_EUNSUPPORTED, "spline_weighting: shape out of range");
    if (E * Mout == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(out && (S * Min == 0 || (x && weight && basis && weight_index)), GNNOPS_EINVAL, "spline_weighting: null pointer");
    const int ochunks = (int)((Mout + 63) / 64);
    const int grid = gnnops_grid_cap(gnnops_cdiv(E * ochunks, 4), 256 * 32);
    hipStream_t stream = (hipStream_t)s;
#define GNNOPS_SW(T)                                                                                                        \
    hipLaunchKernelGGL((spline_weighting_kernel<T>), dim3(grid), dim3(256), 0, stream, (const T*)x, (const T*)weight,       \
                       (const T*)basis, weight_index, (T*)out, E, (int)Min, (int)Mout, (int)S, ochunks)
    switch (dtype) {
        case GNNOPS_F32: GNNOPS_SW(float); break;
        case GNNOPS_F16: GNNOPS_SW(__half); break;
        case GNNOPS_BF16: GNNOPS_SW(__hip_bfloat16); break;
        default: gnnops_set_error("spline_weighting: unknown dtype %d", dtype); return GNNOPS_EINVAL;
    }
#undef GNNOPS_SW
    return gnnops_check_launch("spline_weighting");
}

extern "C" int gnnops_spline_conv(const void* x, const int32_t* rowptr, const int32_t* perm, const int64_t* src, const void* pseudo,
                                  const void* weight, const int64_t* kernel_size, const uint8_t* is_open_spline, int D, int degree,
                                  const void* root_weight, const void* bias, void* out, int64_t N, int64_t E, int64_t Min,
                                  int64_t Mout, int norm, int dtype, gnnops_stream_t s) {
    SplineMeta sm{};
    const int rc = fill_meta(sm, kernel_size, is_open_spline, D, degree, "spline_conv");
    if (rc != GNNOPS_OK) return rc;
    GNNOPS_REQUIRE(N >= 0 && E >= 0 && Min >= 0 && Mout >= 0, GNNOPS_EINVAL, "spline_conv: negative size");
    GNNOPS_REQUIRE(Min < (1 << 20) && Mout < (1 << 20) && E < ((int64_t)1 << 31), GNNOPS_EUNSUPPORTED, "spline_conv: shape out of range");
    if (N * Mout == 0) return GNNOPS_OK;
    GNNOPS_REQUIRE(rowptr && out && (E == 0 || (x && src && pseudo && weight)), GNNOPS_EINVAL, "spline_conv: null pointer");
    hipStream_t stream = (hipStream_t)s;
    switch (dtype) {
        case GNNOPS_F32: return run_conv<float>(x, rowptr, perm, src, pseudo, weight, sm, root_weight, bias, out, N, (int)Min, (int)Mout, norm, stream);
        case GNNOPS_F16: return run_conv<__half>(x, rowptr, perm, src, pseudo, weight, sm, root_weight, bias, out, N, (int)Min, (int)Mout, norm, stream);
        case GNNOPS_BF16: return run_conv<__hip_bfloat16>(x, rowptr, perm, src, pseudo, weight, sm, root_weight, bias, out, N, (int)Min, (int)Mout, norm, stream);
    }
    gnnops_set_error("spline_conv: unknown dtype %d", dtype);
    return GNNOPS_EINVAL;
}
