#include "gemm.cuh"
#include <algorithm>
#include "gemm_api.h"
namespace cc {
int gemm_f32out(int al, int bl, const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, float* C, int ldc,
                const float* bias, int mode, float alpha, int ksplit, hipStream_t st) {
    if ((ldc & 3) || (N & 7)) return CC_ERR_SHAPE;
    if (ksplit > 1 && mode != 2) return CC_ERR_ARG;  // slab mode (3) is reached through gemm_wgrad only
    EpiF32 e{C, bias, ldc, M, N, mode, alpha};
    return launch_gemm(al, bl, A, lda, B, ldb, M, N, K, ksplit, e, st);
}
__global__ __launch_bounds__(256) void k_slab_reduce(const float* __restrict__ slabs, size_t slab_elems, int ks, int Nw, float* __restrict__ dW,
                                                     int ldw, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(slabs)[i];
        for (int s = 1; s < ks; s++) {
            const float4 b = reinterpret_cast<const float4*>(slabs + s * slab_elems)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const size_t e = i * 4;
        float* d = dW + (e / Nw) * ldw + (e % Nw);
        const float4 c = *reinterpret_cast<const float4*>(d);
        *reinterpret_cast<float4*>(d) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
}

int gemm_wgrad(const bf16_t* X, int ldx, const bf16_t* Y, int ldy, int Mw, int Nw, int K, float* dW, int ldw, float* scratch,
               hipStream_t st) {
    if ((Nw & 7) || (ldw & 3)) return CC_ERR_SHAPE;
    const int tiles = ((Mw + G_BM - 1) / G_BM) * ((Nw + G_BN - 1) / G_BN);
    int ks = 512 / (tiles > 0 ? tiles : 1);
    const int kmax = (K + 255) / 256;  // at least 4 K-steps per slice
    if (ks > kmax) ks = kmax;
    const size_t slab = (size_t)Mw * Nw;
    if (scratch) { const size_t fit = WGRAD_SCRATCH_BYTES / (slab * sizeof(float)); if ((size_t)ks > fit) ks = (int)fit; } else ks = 1;
    if (ks <= 1) return gemm_f32out(1, 1, X, ldx, Y, ldy, Mw, Nw, K, dW, ldw, nullptr, 1, 1.0f, 1, st);
    // slices z write slab z (EpiF32 store mode; C pointer advanced per z inside the kernel via blockIdx.z * slab)
    EpiF32 e{scratch, nullptr, Nw, Mw, Nw, 3, 1.0f};
    e.zstride = slab;
    int rc = launch_gemm(1, 1, X, ldx, Y, ldy, Mw, Nw, K, ks, e, st);
    if (rc != CC_OK) return rc;
    // the launcher may have reduced the slice count (ceil division): recompute it the same way
    const int kt = (K + G_BK - 1) / G_BK, per = (kt + ks - 1) / ks;
    const int ks_eff = (kt + per - 1) / per;
    const size_t n4 = slab / 4;
    hipLaunchKernelGGL(k_slab_reduce, dim3((int)std::min<size_t>((n4 + 255) / 256, 2048)), dim3(256), 0, st, scratch, slab, ks_eff, Nw, dW, ldw, n4);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

__global__ __launch_bounds__(256) void k_splitk_finish(const float* __restrict__ slabs, size_t slab_elems, int ks, int M, int N,
                                                       const float* __restrict__ bias, int act, const float* __restrict__ res,
                                                       float* __restrict__ out32, bf16_t* __restrict__ out16, int ldo) {
    const size_t n8 = (size_t)M * (N >> 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / (N >> 3)), col = (int)(i % (N >> 3)) * 8;
        const size_t e = (size_t)row * N + col;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < ks; s++) {
            const float4 a = *reinterpret_cast<const float4*>(slabs + s * slab_elems + e);
            const float4 b = *reinterpret_cast<const float4*>(slabs + s * slab_elems + e + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        if (bias) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] += bias[col + k];
        }
        if (act == 1) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = fmaxf(v[k], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = gelu_new_f(v[k]);
        }
        const size_t o = (size_t)row * ldo + col;
        if (res) {
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] += res[o + k];
        }
        if (out32) {
            *reinterpret_cast<float4*>(out32 + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(out32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (out16) *reinterpret_cast<uint4*>(out16 + o) = pack8(v);
    }
}

int gemm_nt_skinny(const bf16_t* A, int lda, const bf16_t* B, int ldb, int M, int N, int K, const float* bias, int act, const float* res,
                   float* out32, bf16_t* out16, int ldo, float* scratch, size_t scratch_bytes, hipStream_t st) {
    if ((N & 7) || (ldo & 7)) return CC_ERR_SHAPE;
    const int tiles = ((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN);
    int ks = 384 / (tiles > 0 ? tiles : 1);
    const int kmax = K / (2 * G_BK);                       // at least 2 K-steps per slice
    if (ks > kmax) ks = kmax;
    const size_t slab = (size_t)M * N;
    if (scratch && slab && (K % G_BK) == 0) { const size_t fit = scratch_bytes / (slab * sizeof(float)); if ((size_t)ks > fit) ks = (int)fit; } else ks = 1;
    if (ks <= 1) {
        if (res) { EpiResid e{out32, res, bias, ldo, M, N}; return launch_gemm(0, 0, A, lda, B, ldb, M, N, K, 1, e, st); }
        if (out16) { EpiBF16 e{out16, nullptr, bias, ldo, M, N, act}; return launch_gemm(0, 0, A, lda, B, ldb, M, N, K, 1, e, st); }
        EpiF32 e{out32, bias, ldo, M, N, 0, 1.0f};
        return launch_gemm(0, 0, A, lda, B, ldb, M, N, K, 1, e, st);
    }
    EpiF32 e{scratch, nullptr, N, M, N, 3, 1.0f};
    e.zstride = slab;
    int rc = launch_gemm(0, 0, A, lda, B, ldb, M, N, K, ks, e, st);
    if (rc != CC_OK) return rc;
    const int kt = K / G_BK, per = (kt + ks - 1) / ks, ks_eff = (kt + per - 1) / per;
    const size_t n8 = (size_t)M * (N >> 3);
    hipLaunchKernelGGL(k_splitk_finish, dim3((int)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, st, scratch, slab, ks_eff, M, N, bias, act,
                       res, out32, out16, ldo);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}
}  // namespace cc
