// bf16 MFMA GEMM for gfx950 (MI355X): C[M,N] = A · B with fp32 accumulation and a fused, vectorised epilogue.
//
// One kernel template covers the three operand layouts the ClipCap path needs, so that neither weights nor
// activations are ever transposed in HBM:
//   AL=0: A stored [M][K] (K contiguous)        AL=1: A stored [K][M] (M contiguous; wgrad: dY^T)
//   BL=0: B stored [N][K] (torch.nn.Linear W)   BL=1: B stored [K][N] (HF Conv1D W, dgrad of Linear, wgrad: X)
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16.
// LDS image is always [row][k] (128 B rows, 16-B chunks XOR-swizzled) so fragments are single ds_read_b128:
//   chunk' = chunk ^ ((row>>1)&7) ^ ((row>>4)&3)
// -> ds_read_b128 fragment reads, ds_write_b128 (K-contiguous staging) and ds_write_b64 (transposing staging) are all
// bank-conflict free (derivation in DESIGN.md §GEMM).  K-strided operands are transposed in registers on the way
// to LDS (4 k-rows x 8 columns per thread, v_perm_b32 16-bit interleave).  Global->register loads for tile t+1
// are issued before the MFMAs of tile t and written to the other LDS buffer after them (one barrier per K-step).
// The epilogue transposes each wave's accumulators through a wave-private LDS strip so that every lane owns 8
// consecutive columns of one row: bias/residual/activation loads and the C stores are 16-B/32-B vectors.
//
// Constraints (checked by the host launcher): leading dims and the contiguous extent of every operand are
// multiples of 8 elements, base pointers 16-B aligned.  M, N (row counts) and K of a K-strided operand are free.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.hip.h"
#include "gemm_api.h"

namespace CC_NS {

constexpr int G_BM = 128, G_BN = 128, G_BK = 64, G_THREADS = 256;
constexpr int G_TILE_BYTES = G_BM * G_BK * 2;  // 16 KiB per operand per buffer
constexpr int G_EPI_LD = 68;                    // floats per row of the epilogue strip (64 + pad, keeps 16-B alignment)

struct GemmShape {
    int M, N, K;
    int lda, ldb;
    int k_chunk;  // K extent handled by one blockIdx.z slice (multiple of 64); == K rounded up when no split
    int group_m;  // m-tiles per band of the tile order (tile_coords)
    int stagger;  // 128 x 128 two-blocks-per-CU kernel: s_sleep units (64 cycles) the second block of every CU waits before its first tile
    int flags = 0;  // bit 0 (lab A/B, CC_Q4_LAST=0): the 4-wave kernel keeps its dummy tail stream on a workgroup's last tile
};

__device__ __forceinline__ int g_lds_off(int row, int chunk) {
    return row * 128 + (((chunk ^ (row >> 1) ^ ((row >> 4) & 3)) & 7) << 4);
}

struct StageRegs {
    uint4 v[4];
};

// ---- K-contiguous operand: global [rows][ld]; thread -> (chunk = tid&7, rows tid>>3 + 32 i) ----
__device__ __forceinline__ void g_load_kc(StageRegs& s, const op16_t* __restrict__ base, int ld, int rows, int r0,
                                          int k0, int kend, int tid) {
    const int c = tid & 7, rr = tid >> 3;
    const int k = k0 + c * 8;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int r = r0 + rr + 32 * i;
        if (r < rows && k < kend)
            s.v[i] = *reinterpret_cast<const uint4*>(base + (size_t)r * ld + k);
        else
            s.v[i] = make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void l_store_kc(const StageRegs& s, char* lds, int tid) {
    const int c = tid & 7, rr = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; i++) *reinterpret_cast<uint4*>(lds + g_lds_off(rr + 32 * i, c)) = s.v[i];
}

// ---- K-strided operand: global [k][ld] with the tile's rows contiguous; thread -> 4 k-rows x 8 rows ----
__device__ __forceinline__ void g_load_ks(StageRegs& s, const op16_t* __restrict__ base, int ld, int rows, int r0,
                                          int k0, int kend, int tid) {
    const int ng = (tid & 7) | (((tid >> 4) & 1) << 3);
    const int kg = ((tid >> 3) & 1) | ((tid >> 5) << 1);
    const int r = r0 + ng * 8;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = k0 + kg * 4 + j;
        if (k < kend && r < rows)
            s.v[j] = *reinterpret_cast<const uint4*>(base + (size_t)k * ld + r);
        else
            s.v[j] = make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void l_store_ks(const StageRegs& s, char* lds, int tid) {
    const int ng = (tid & 7) | (((tid >> 4) & 1) << 3);
    const int kg = ((tid >> 3) & 1) | ((tid >> 5) << 1);
    const int kc = kg >> 1, half = (kg & 1) * 8;
    const unsigned in[4][4] = {{s.v[0].x, s.v[0].y, s.v[0].z, s.v[0].w},
                               {s.v[1].x, s.v[1].y, s.v[1].z, s.v[1].w},
                               {s.v[2].x, s.v[2].y, s.v[2].z, s.v[2].w},
                               {s.v[3].x, s.v[3].y, s.v[3].z, s.v[3].w}};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        // even column: low halves of the 4 k-rows; odd column: high halves
        uint2 ev, od;
        ev.x = __builtin_amdgcn_perm(in[1][c], in[0][c], 0x05040100u);
        ev.y = __builtin_amdgcn_perm(in[3][c], in[2][c], 0x05040100u);
        od.x = __builtin_amdgcn_perm(in[1][c], in[0][c], 0x07060302u);
        od.y = __builtin_amdgcn_perm(in[3][c], in[2][c], 0x07060302u);
        const int n = ng * 8 + 2 * c;
        *reinterpret_cast<uint2*>(lds + g_lds_off(n, kc) + half) = ev;
        *reinterpret_cast<uint2*>(lds + g_lds_off(n + 1, kc) + half) = od;
    }
}

template <int L>
__device__ __forceinline__ void g_load(StageRegs& s, const op16_t* base, int ld, int rows, int r0, int k0, int kend,
                                       int tid) {
    if constexpr (L == 0)
        g_load_kc(s, base, ld, rows, r0, k0, kend, tid);
    else
        g_load_ks(s, base, ld, rows, r0, k0, kend, tid);
}
template <int L>
__device__ __forceinline__ void l_store(const StageRegs& s, char* lds, int tid) {
    if constexpr (L == 0)
        l_store_kc(s, lds, tid);
    else
        l_store_ks(s, lds, tid);
}

// xcd_remap (XCD-aware block order) lives in common.hip.h

// Functors whose epilogue reads a second tensor (dgrad-through-activation: the activation input) can hand it over in the strip
// layout before the first store: Epi::StripAux / load_aux(row, col) / operator()(row, col, v, aux).  Otherwise every such load sits
// between two stores, and with loads and stores on the one vmcnt counter the compiler waits vmcnt(0) for it — a store-queue drain
// per 16-row strip.
// epilogue functors that reduce along a row (lm_head softmax partials) take the wave's whole 64-column strip: Epi::strip()
template <class E> struct epi_row_strip { static constexpr bool value = false; };
// functors whose additive input can initialise the accumulators (Epi::init_from_input() / init4(row, col, acc4)): see gemm_stag256_body
template <class E, class = void> struct epi_acc_init { static constexpr bool value = false; };
template <class E> struct epi_acc_init<E, decltype((void)&E::init4)> { static constexpr bool value = true; };
template <class E, class = void> struct epi_strip_aux { static constexpr bool value = false; };
template <class E> struct epi_strip_aux<E, decltype((void)sizeof(typename E::StripAux))> { static constexpr bool value = true; };

// ---- epilogue: wave-private LDS strip [16][68] fp32; C layout of 16x16x32: col = lane&15, row = (lane>>4)*4 + reg.
// Every lane ends up with 8 consecutive columns of one row -> vector epilogue.  Caller must have passed a barrier
// after the last LDS read of the main loop.
template <class Epi, int I0 = 0, int I1 = 4>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[4][4], char* smem, int wave, int lane, int row0, int col0, const Epi& epi) {
    float* strip = reinterpret_cast<float*>(smem) + wave * (16 * G_EPI_LD);
    const int er = (lane >> 4) * 4, ec = lane & 15;
    if constexpr (epi_strip_aux<Epi>::value) {
        typename Epi::StripAux aux[4][2];
#pragma unroll
        for (int i = I0; i < I1; i++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int q = lane + 64 * s;
                aux[i][s] = epi.load_aux(row0 + i * 16 + (q >> 3), col0 + (q & 7) * 8);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = I0; i < I1; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) strip[(er + r) * G_EPI_LD + j * 16 + ec] = acc[i][j][r];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int q = lane + 64 * s;
                const int lr = q >> 3, c8 = q & 7;
                float v[8];
                const float4 a = *reinterpret_cast<const float4*>(strip + lr * G_EPI_LD + c8 * 8);
                const float4 b = *reinterpret_cast<const float4*>(strip + lr * G_EPI_LD + c8 * 8 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                epi(row0 + i * 16 + lr, col0 + c8 * 8, v, aux[i][s]);
            }
        }
        return;
    }
#pragma unroll
    for (int i = I0; i < I1; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) strip[(er + r) * G_EPI_LD + j * 16 + ec] = acc[i][j][r];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int q = lane + 64 * s;
            const int lr = q >> 3, c8 = q & 7;
            float v[8];
            const float4 a = *reinterpret_cast<const float4*>(strip + lr * G_EPI_LD + c8 * 8);
            const float4 b = *reinterpret_cast<const float4*>(strip + lr * G_EPI_LD + c8 * 8 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            epi(row0 + i * 16 + lr, col0 + c8 * 8, v);
        }
    }
}

// ---- epilogue straight from registers (no LDS) for the 256-row kernels (8 waves, one block per CU: nothing else on the CU hides
// an LDS round trip and its barrier; with the 128 x 128 kernels' two co-resident blocks the strip epilogue above measured 1.5 %
// faster on the training step).  The kernel issues its MFMAs with the operands SWAPPED (B fragment first), so
// a lane holds C[row = lane & 15][cols 4 q .. 4 q + 3] of each 16 x 16 tile (q = lane >> 4): four CONSECUTIVE columns of one row.
// v_permlane16_swap on the accumulators of two tiles X, Y hands the odd 16-lane rows of X to Y and the even rows of Y to X;
// afterwards lane q holds eight consecutive columns 8 (q >> 1) .. + 7 of tile (q & 1 ? Y : X) — the functors' (row, col, v[8])
// unit.  Tiles pair along j; a leftover column tile (odd NJ) pairs along i.
// Order: every global LOAD of the epilogue (residual / activation input / bias / targets) is issued before its first STORE: with
// loads and stores both pending on the single vmcnt counter the compiler has to wait vmcnt(0) — drain the store queue — before
// it may use a loaded value, which serialised the old load-store-load-store epilogue on the HBM write latency.
//   Epi::pre4(row, col, acc4)        loads folded into one tile's accumulators (accumulator layout), kPre only
//   Epi::bias8(col, b[8])            the lane's bias values for a column group
//   Epi::fin(row, col, v[8], b[8])   math + stores only
//   Epi::strip(...)                  row-strip functors (NJ == 4): both column groups of the row at once
template <class Epi, int NI, int NJ>
__device__ __forceinline__ void gemm_epilogue_regs(f32x4 (&acc)[NI][NJ], int lane, int row0, int col0, const Epi& epi) {
    static_assert(NI % 2 == 0 || NJ % 2 == 0, "leftover column tiles pair along i");
    const int q = lane >> 4, rl = lane & 15;
#define G_PAIR(X, Y, V)                                                                                                  \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; r_++) {                                                                   \
        const auto s_ = __builtin_amdgcn_permlane16_swap(__float_as_uint((X)[r_]), __float_as_uint((Y)[r_]), false, false); \
        (V)[r_] = __uint_as_float(s_[0]);                                                                                \
        (V)[4 + r_] = __uint_as_float(s_[1]);                                                                            \
    }
    if constexpr (epi_row_strip<Epi>::value) {
        static_assert(!epi_row_strip<Epi>::value || NJ == 4, "row-strip epilogues need a 64-column wave strip");
        typename Epi::RowAux tg[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) tg[i] = epi.load_row(row0 + i * 16 + rl);
#pragma unroll
        for (int i = 0; i < NI; i++) {
            float va[8], vb[8];
            G_PAIR(acc[i][0], acc[i][1], va);
            G_PAIR(acc[i][NJ - 2], acc[i][NJ - 1], vb);
            const int cb = col0 + (q & 1) * 16 + (q >> 1) * 8;
            epi.strip(row0 + i * 16 + rl, cb, cb + 32, va, vb, tg[i]);
        }
    } else {
        if constexpr (Epi::kPre) {
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) epi.pre4(row0 + i * 16 + rl, col0 + j * 16 + q * 4, acc[i][j]);
        }
        float bias[(NJ + 1) / 2][8];
#pragma unroll
        for (int jp = 0; jp < NJ / 2; jp++) epi.bias8(col0 + (2 * jp + (q & 1)) * 16 + (q >> 1) * 8, bias[jp]);
        if constexpr (NJ & 1) epi.bias8(col0 + (NJ - 1) * 16 + (q >> 1) * 8, bias[NJ / 2]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; i++) {
#pragma unroll
            for (int jp = 0; jp < NJ / 2; jp++) {
                float v[8];
                G_PAIR(acc[i][2 * jp], acc[i][2 * jp + 1], v);
                epi.fin(row0 + i * 16 + rl, col0 + (2 * jp + (q & 1)) * 16 + (q >> 1) * 8, v, bias[jp]);
            }
            if constexpr (NJ & 1) {
                if (i & 1) {
                    float v[8];
                    G_PAIR(acc[i - 1][NJ - 1], acc[i][NJ - 1], v);
                    epi.fin(row0 + (i - 1 + (q & 1)) * 16 + rl, col0 + (NJ - 1) * 16 + (q >> 1) * 8, v, bias[NJ / 2]);
                }
            }
        }
    }
#undef G_PAIR
}

// tile id -> (m-tile, n-tile): ids walk GROUP_M m-tiles down, then one n-tile across (column-major inside a band of
// GROUP_M m-tiles).  With the XCD remap above, the ~64 blocks resident on one XCD cover an 8x8 patch of tiles, whose
// A and B panels (16 x 128 x K bf16) fit that XCD's 4 MiB L2 instead of streaming B once per m-tile.
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int G, int& tm, int& tn) {
    const int band = tile / (G * tiles_n);
    const int first = band * G;
    const int gm = min(G, tiles_m - first);
    const int r = tile - band * G * tiles_n;
    tm = first + r % gm;
    tn = r / gm;
}

template <int AL, int BL, class Epi>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_bf16_kernel(const op16_t* __restrict__ A,
                                                                  const op16_t* __restrict__ B, GemmShape g, Epi epi) {
    __shared__ __attribute__((aligned(16))) char smem[4 * G_TILE_BYTES];  // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);
    const int nk = (kend - kbeg + G_BK - 1) / G_BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    StageRegs ra, rb;
    if (nk > 0) {
        g_load<AL>(ra, A, g.lda, g.M, m0, kbeg, kend, tid);
        g_load<BL>(rb, B, g.ldb, g.N, n0, kbeg, kend, tid);
        l_store<AL>(ra, smem, tid);
        l_store<BL>(rb, smem + G_TILE_BYTES, tid);
    }
    __syncthreads();

    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        char* cur = smem + (kt & 1) * 2 * G_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * G_TILE_BYTES;
        const bool more = (kt + 1 < nk);
        if (more) {
            const int k0 = kbeg + (kt + 1) * G_BK;
            g_load<AL>(ra, A, g.lda, g.M, m0, k0, kend, tid);
            g_load<BL>(rb, B, g.ldb, g.N, n0, k0, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            op16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                af[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 64 + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < 4; j++)
                bfr[j] = *reinterpret_cast<const op16x8*>(cur + G_TILE_BYTES +
                                                          g_lds_off(wn * 64 + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[i][j] = CC_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
        }
        if (more) {
            l_store<AL>(ra, nxt, tid);
            l_store<BL>(rb, nxt + G_TILE_BYTES, tid);
        }
        __syncthreads();
    }

    gemm_epilogue(acc, smem, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
}

// ------------------------------------------------------------------------------------------------
// NT fast path: both operands K-contiguous (A [M][K], B [N][K]), K % 64 == 0.  Tiles go HBM -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write): the LDS destination of a wave instruction is
// wave-uniform base + lane*16, so the XOR swizzle is applied on the per-lane SOURCE address (chunk ^ f(row)) and the
// same involution on the fragment read (cdna guide rule 21).  Rows beyond M/N are clamped (their results are
// dropped by the epilogue).  One barrier per K-step; the loads of tile t+1 are in flight during the MFMAs of tile t.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds_tile(const op16_t* __restrict__ base, int ld, int rows, int r0, int k0, char* lds, int wave, int lane) {
    // the tile is 16 wave-segments of 1 KiB (8 rows x 128 B); wave w fills segments w, w+4, w+8, w+12
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int seg = wave + 4 * i;
        const int row = seg * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7) ^ ((row >> 4) & 3);
        const int gr = min(r0 + row, rows - 1);
        const op16_t* src = base + (size_t)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + seg * 1024), 16, 0, 0);
    }
}

// ABL: ablation bits for tools/gemm_bench.py (0 in every product instantiation): 1 = no loads in the K loop, 2 = no MFMAs,
// 4 = no fragment reads (results are then meaningless; timing only)
template <class Epi, int ABL = 0>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_nt_glds_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B,
                                                                     GemmShape g, Epi epi) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * G_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_chunk;                  // split-K slice (k_chunk is a multiple of 64)
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Two blocks share a CU.  Launched together they run in lockstep — both in the MFMA loop, then both in the (VALU) epilogue — so the
    // two pipes never overlap.  The blocks of the odd 256-block dispatch waves start a fraction of a tile late, which keeps one block's
    // epilogue under the other's main loop for the rest of the launch.
    if (g.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
        for (int s_ = g.stagger; s_ > 0; s_ -= 64) __builtin_amdgcn_s_sleep(64);
    }
    glds_tile(A, g.lda, g.M, m0, kbeg, smem, wave, lane);
    glds_tile(B, g.ldb, g.N, n0, kbeg, smem + G_TILE_BYTES, wave, lane);
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        char* cur = smem + (kt & 1) * 2 * G_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * G_TILE_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk && !(ABL & 1)) {
            glds_tile(A, g.lda, g.M, m0, kbeg + (kt + 1) * G_BK, nxt, wave, lane);
            glds_tile(B, g.ldb, g.N, n0, kbeg + (kt + 1) * G_BK, nxt + G_TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            op16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (!(ABL & 4) || kt == 0) af[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 64 + i * 16 + frow, ks * 4 + fchunk));
                else asm volatile("" : "=v"(af[i]));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!(ABL & 4) || kt == 0) bfr[j] = *reinterpret_cast<const op16x8*>(cur + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, ks * 4 + fchunk));
                else asm volatile("" : "=v"(bfr[j]));
            }
            if (ABL & 2) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("" ::"v"(af[i]), "v"(bfr[i]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i][j] = CC_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
            }
        }
    }
    __syncthreads();
    gemm_epilogue(acc, smem, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
}

// bf16x3 build: the same kernel in the fused two-stage form of gemm_stag256_body<X3F> — a K chunk of 64 is its hi tiles
// (A image columns [c, c + 64), B image columns [c, c + 64)), then its lo tiles (A: 2 K + c, B: K + c); the hi stage's fragments stay in
// registers (64 of them: two 32-deep sub-steps x (4 + 4) fragments) and the lo stage runs A_hi B_lo + A_lo B_hi.  g.K is K' = 3 K, one slice.
template <class Epi>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_nt_glds_x3f_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B,
                                                                         GemmShape g, Epi epi) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * G_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int Kl = g.K / 3, nc = Kl / G_BK;                   // logical K, chunks
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15, fchunk = lane >> 4;
    op16x8 ah[2][4], bh[2][4];
#define X3F_ISSUE(C, LO, BUF)                                                                                            \
    {                                                                                                                    \
        glds_tile(A, g.lda, g.M, m0, (C)*G_BK + ((LO) ? 2 * Kl : 0), (BUF), wave, lane);                                 \
        glds_tile(B, g.ldb, g.N, n0, (C)*G_BK + ((LO) ? Kl : 0), (BUF) + G_TILE_BYTES, wave, lane);                      \
    }
    char* buf0 = smem;
    char* buf1 = smem + 2 * G_TILE_BYTES;
    X3F_ISSUE(0, 0, buf0)
    for (int c = 0; c < nc; c++) {
        // ---- hi stage (buffer 0); the lo tiles of this chunk travel meanwhile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        X3F_ISSUE(c, 1, buf1)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
#pragma unroll
            for (int i = 0; i < 4; i++) ah[ks][i] = *reinterpret_cast<const op16x8*>(buf0 + g_lds_off(wm * 64 + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < 4; j++) bh[ks][j] = *reinterpret_cast<const op16x8*>(buf0 + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(ah[ks][i], bh[ks][j], acc[i][j]);
        }
        // ---- lo stage (buffer 1); the next chunk's hi tiles travel meanwhile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < nc) X3F_ISSUE(c + 1, 0, buf0)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            op16x8 al[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; i++) al[i] = *reinterpret_cast<const op16x8*>(buf1 + g_lds_off(wm * 64 + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < 4; j++) bl[j] = *reinterpret_cast<const op16x8*>(buf1 + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(ah[ks][i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(al[i], bh[ks][j], acc[i][j]);
        }
    }
#undef X3F_ISSUE
    __syncthreads();
    gemm_epilogue(acc, smem, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
}

// Small grids (<= one 128 x 128 tile per CU): the same kernel with FOUR LDS stages (128 KiB, one block per CU by construction) and
// three K-tiles of DMA in flight behind a counted vmcnt.  With a single block per CU the 2-stage kernel above exposes a full
// L2/HBM round trip per K-step (s_waitcnt vmcnt(0) with nothing else on the CU to run): mapper GEMMs (M = 5120) ran 12-24 K-steps
// at ~0.85 us each.  On grids that fill the CUs twice the 2-stage kernel with two co-resident blocks stays better (measured -30 %
// for this variant there), so launch_gemm only picks it when tiles <= CUs.
template <class Epi>
__global__ __launch_bounds__(G_THREADS, 1) void gemm_nt_glds4_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B,
                                                                      GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smem4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define G4_ISSUE(T)                                                                                                      \
    {                                                                                                                    \
        char* st_ = smem4 + ((T) & 3) * 2 * G_TILE_BYTES;                                                                \
        glds_tile(A, g.lda, g.M, m0, kbeg + (T)*G_BK, st_, wave, lane);                                                  \
        glds_tile(B, g.ldb, g.N, n0, kbeg + (T)*G_BK, st_ + G_TILE_BYTES, wave, lane);                                   \
    }
    G4_ISSUE(0);
    if (nk > 1) G4_ISSUE(1);
    if (nk > 2) G4_ISSUE(2);
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        // tile kt must have landed; tiles kt+1, kt+2 (8 DMA instructions per wave each) may stay in flight
        const int rem = min(nk - 1, kt + 2) - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // raw barrier: __syncthreads() would drain the DMA queue
        if (kt + 3 < nk) G4_ISSUE(kt + 3);             // buffer (kt+3)&3 held tile kt-1, which every wave finished before this barrier
        const char* cur = smem4 + (kt & 3) * 2 * G_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            op16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) af[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 64 + i * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int j = 0; j < 4; j++) bfr[j] = *reinterpret_cast<const op16x8*>(cur + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, ks * 4 + fchunk));
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
        }
    }
#undef G4_ISSUE
    __syncthreads();
    gemm_epilogue(acc, smem4, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
}

// Small-grid kernel, 8-wave form: the same 128 x 128 tile and 4-stage DMA pipeline, but the two 32-k halves of every K-tile go to
// two wave groups (waves 0-3 / 4-7, same 64 x 64 wave tiles).  With one block per CU the 4-wave kernel has ONE wave per SIMD, so
// every fragment read and MFMA chain is exposed; here each SIMD holds two waves with half the per-tile work each, and the DMA issue
// (4 instead of 8 instructions per wave and tile pair) is spread over twice the waves.  The groups' partial sums are exchanged
// through LDS after the loop (group 1 hands over row strips 0-1, group 0 strips 2-3) and both groups run half of the epilogue.
template <class Epi>
__global__ __launch_bounds__(2 * G_THREADS, 1) void gemm_nt_glds4x2_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B,
                                                                            GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smem4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // group 0 streams the A tile, group 1 the B tile: 4 DMA instructions per wave and K-tile
#define G4_ISSUE(T)                                                                                                      \
    {                                                                                                                    \
        char* st_ = smem4 + ((T) & 3) * 2 * G_TILE_BYTES;                                                                \
        if (grp == 0) glds_tile(A, g.lda, g.M, m0, kbeg + (T)*G_BK, st_, w4, lane);                                      \
        else glds_tile(B, g.ldb, g.N, n0, kbeg + (T)*G_BK, st_ + G_TILE_BYTES, w4, lane);                                \
    }
    G4_ISSUE(0);
    if (nk > 1) G4_ISSUE(1);
    if (nk > 2) G4_ISSUE(2);
    const int frow = lane & 15, fchunk = lane >> 4;
    for (int kt = 0; kt < nk; kt++) {
        const int rem = min(nk - 1, kt + 2) - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < nk) G4_ISSUE(kt + 3);
        const char* cur = smem4 + (kt & 3) * 2 * G_TILE_BYTES;
        op16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 64 + i * 16 + frow, grp * 4 + fchunk));
#pragma unroll
        for (int j = 0; j < 4; j++) bfr[j] = *reinterpret_cast<const op16x8*>(cur + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, grp * 4 + fchunk));
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
    }
#undef G4_ISSUE
    __syncthreads();
    // exchange: 8 float4 per thread each way, [tile][thread] so that a wave's ds_write/read_b128 is contiguous
    f32x4* xch = reinterpret_cast<f32x4*>(smem4 + 4 * 2 * G_TILE_BYTES / 2) + grp * (8 * G_THREADS) + (tid & (G_THREADS - 1));
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) xch[(i * 4 + j) * G_THREADS] = grp ? acc[i][j] : acc[2 + i][j];
    __syncthreads();
    const f32x4* xin = reinterpret_cast<const f32x4*>(smem4 + 4 * 2 * G_TILE_BYTES / 2) + (grp ^ 1) * (8 * G_THREADS) + (tid & (G_THREADS - 1));
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] += xin[(i * 4 + j) * G_THREADS];
        gemm_epilogue<Epi, 0, 2>(acc, smem4, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
    } else {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[2 + i][j] += xin[(i * 4 + j) * G_THREADS];
        gemm_epilogue<Epi, 2, 4>(acc, smem4, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
    }
}

// bf16x3 build: the 8-wave small-grid kernel in the fused two-stage form (gemm_stag256_body<X3F>): hi stage, lo stage, hi fragments retained
template <class Epi>
__global__ __launch_bounds__(2 * G_THREADS, 1) void gemm_nt_glds4x2_x3f_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B,
                                                                            GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smem4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int Kl = g.K / 3;                                   // logical K (g.K = K' = 3 K, one slice)
    const int nk = 2 * (Kl / G_BK);                           // a K chunk = its hi stage, then its lo stage
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // group 0 streams the A tile, group 1 the B tile: 4 DMA instructions per wave and K-tile
#define G4_ISSUE(T)                                                                                                      \
    {                                                                                                                    \
        char* st_ = smem4 + ((T) & 3) * 2 * G_TILE_BYTES;                                                                \
        if (grp == 0) glds_tile(A, g.lda, g.M, m0, ((T) >> 1) * G_BK + (((T) & 1) ? 2 * Kl : 0), st_, w4, lane);           \
        else glds_tile(B, g.ldb, g.N, n0, ((T) >> 1) * G_BK + (((T) & 1) ? Kl : 0), st_ + G_TILE_BYTES, w4, lane);       \
    }
    G4_ISSUE(0);
    if (nk > 1) G4_ISSUE(1);
    if (nk > 2) G4_ISSUE(2);
    const int frow = lane & 15, fchunk = lane >> 4;
#define G4_WAIT(KT)                                                                                                      \
    {                                                                                                                    \
        const int rem_ = min(nk - 1, (KT) + 2) - (KT);                                                                   \
        if (rem_ >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                  \
        else if (rem_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                             \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                            \
        __builtin_amdgcn_s_barrier();                                                                                    \
        if ((KT) + 3 < nk) G4_ISSUE((KT) + 3);                                                                           \
    }
    op16x8 af[4], bfr[4];
    for (int kt = 0; kt < nk; kt += 2) {
        G4_WAIT(kt)
        const char* cur = smem4 + (kt & 3) * 2 * G_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; i++) af[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 64 + i * 16 + frow, grp * 4 + fchunk));
#pragma unroll
        for (int j = 0; j < 4; j++) bfr[j] = *reinterpret_cast<const op16x8*>(cur + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, grp * 4 + fchunk));
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(af[i], bfr[j], acc[i][j]);
        G4_WAIT(kt + 1)
        const char* cur2 = smem4 + ((kt + 1) & 3) * 2 * G_TILE_BYTES;
        op16x8 al[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; i++) al[i] = *reinterpret_cast<const op16x8*>(cur2 + g_lds_off(wm * 64 + i * 16 + frow, grp * 4 + fchunk));
#pragma unroll
        for (int j = 0; j < 4; j++) bl[j] = *reinterpret_cast<const op16x8*>(cur2 + G_TILE_BYTES + g_lds_off(wn * 64 + j * 16 + frow, grp * 4 + fchunk));
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(af[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = CC_MFMA_16x16x32(al[i], bfr[j], acc[i][j]);
    }
#undef G4_WAIT
#undef G4_ISSUE
    __syncthreads();
    // exchange: 8 float4 per thread each way, [tile][thread] so that a wave's ds_write/read_b128 is contiguous
    f32x4* xch = reinterpret_cast<f32x4*>(smem4 + 4 * 2 * G_TILE_BYTES / 2) + grp * (8 * G_THREADS) + (tid & (G_THREADS - 1));
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) xch[(i * 4 + j) * G_THREADS] = grp ? acc[i][j] : acc[2 + i][j];
    __syncthreads();
    const f32x4* xin = reinterpret_cast<const f32x4*>(smem4 + 4 * 2 * G_TILE_BYTES / 2) + (grp ^ 1) * (8 * G_THREADS) + (tid & (G_THREADS - 1));
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] += xin[(i * 4 + j) * G_THREADS];
        gemm_epilogue<Epi, 0, 2>(acc, smem4, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
    } else {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[2 + i][j] += xin[(i * 4 + j) * G_THREADS];
        gemm_epilogue<Epi, 2, 4>(acc, smem4, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
    }
}

// ------------------------------------------------------------------------------------------------
// Skinny NT kernel for decode-sized M (a few hundred rows): block tile 64 x (64 NJ), 4 waves as 2 x 2, wave tile 32 x 32 NJ on
// v_mfma_f32_32x32x16_bf16.  Why not the 128 x 128 kernels: at M = 320 they give 24-96 blocks, and ONE block's K-step is bound by
// what a single CU can move — 32 KiB L2->LDS (64 B/clk) plus 64 KiB of fragment reads through its LDS (128 B/clk) ~ 1200 clk per
// 64-k step, independent of M (measured 0.59 us per step for M = 64 ... 320).  Here a K-step moves 16 KiB (NJ = 1) into LDS and
// reads 16 KiB of fragments (a 32 x 32 x 16 MFMA needs half the operand bytes per flop of the 16 x 16 x 32 form), M = 320 is
// exactly 5 row tiles, and the grid has 2.5-5x the blocks, two of them co-resident per CU.  Same LDS image as glds_tile (128-B
// rows, the XOR key is also conflict-free for the 32-row x 2-chunk fragment pattern), NS-stage DMA pipeline with counted vmcnt.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int ROWS>
__device__ __forceinline__ void glds_rows(const op16_t* __restrict__ base, int ld, int rows, int r0, int k0, char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; i++) {                   // ROWS / 8 segments of 1 KiB (8 rows x 128 B); wave w takes w, w+4, ...
        const int seg = wave + 4 * i;
        const int row = seg * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7) ^ ((row >> 4) & 3);
        const int gr = min(r0 + row, rows - 1);
        const op16_t* src = base + (size_t)gr * ld + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + seg * 1024), 16, 0, 0);
    }
}
template <int N> __device__ __forceinline__ void s_wait_vm() {
    // s_waitcnt vmcnt(N) only: gfx9 encoding vmcnt = imm[3:0] | imm[15:14], expcnt imm[6:4] and lgkmcnt imm[11:8] left at "don't wait"
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
    asm volatile("" ::: "memory");
}
// KG = 2: eight waves, the two 32-k halves of a K-tile go to two wave groups (same 2 x 2 wave tiles), which halves each wave's
// serial chain per K-step (DMA issue, fragment reads, MFMAs); the groups swap half of their partial sums through LDS at the end and
// each finishes 16 of the wave tile's 32 rows.
template <int ROWS_A, int ROWS_B, int W>
__device__ __forceinline__ void glds_pair(const op16_t* __restrict__ A, int lda, int M, int m0, const op16_t* __restrict__ B, int ldb, int N, int n0,
                                          int k0, char* lds, int wave, int lane) {
    // (ROWS_A + ROWS_B) / 8 segments of 1 KiB (8 rows x 128 B), A's first; wave w of W takes w, w + W, ...
#pragma unroll
    for (int i = 0; i < (ROWS_A + ROWS_B) / 8 / W; i++) {
        const bool is_a = W * i < ROWS_A / 8;                 // compile-time: W divides ROWS_A / 8
        const int seg = wave + W * i - (is_a ? 0 : ROWS_A / 8);
        const int row = seg * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7) ^ ((row >> 4) & 3);
        const op16_t* src = is_a ? A + (size_t)min(m0 + row, M - 1) * lda + k0 + chunk * 8 : B + (size_t)min(n0 + row, N - 1) * ldb + k0 + chunk * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + (is_a ? 0 : ROWS_A * 128) + seg * 1024), 16, 0, 0);
    }
}
template <class Epi, int NJ, int NS, int KG>
__global__ __launch_bounds__(G_THREADS * KG, (NS * (64 + 64 * NJ) * 128 <= 80 * 1024) ? 2 : 1) void gemm_nt_s64_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B, GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char sm64[];
    constexpr int BN = 64 * NJ, STAGE = (64 + BN) * 128, W = 4 * KG, P = (8 + 8 * NJ) / W;    // P: DMA instructions per wave and K-tile
    static_assert(NS >= 3 && NS <= 8 && (KG == 1 || KG == 2) && (NS - 2) * P < 64, "stage count / groups");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + 63) / 64;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * 64, n0 = tn * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
#define S64_ISSUE(T, SLOT) glds_pair<64, BN, W>(A, g.lda, g.M, m0, B, g.ldb, g.N, n0, kbeg + (T)*G_BK, sm64 + (SLOT) * STAGE, wave, lane)
#pragma unroll
    for (int t = 0; t < NS - 1; t++)
        if (t < nk) S64_ISSUE(t, t);
    const int frow = lane & 31, fhalf = lane >> 5;
    int slot = 0, islot = NS - 1;
    for (int kt = 0; kt < nk; kt++) {
        // tile kt must have landed; up to NS-2 later tiles (P DMA instructions per wave each) stay in flight
        const int rem = min(nk - 1, kt + NS - 2) - kt;
        if (rem >= 6) s_wait_vm<(NS > 7 ? 6 : 0) * P>();
        else if (rem == 5) s_wait_vm<(NS > 6 ? 5 : 0) * P>();
        else if (rem == 4) s_wait_vm<(NS > 5 ? 4 : 0) * P>();
        else if (rem == 3) s_wait_vm<(NS > 4 ? 3 : 0) * P>();
        else if (rem == 2) s_wait_vm<(NS > 3 ? 2 : 0) * P>();
        else if (rem == 1) s_wait_vm<P>();
        else s_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) S64_ISSUE(kt + NS - 1, islot);      // that slot held tile kt-1: every wave is past it
        const char* cur = sm64 + slot * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4 / KG; kk++) {
            const int ch = (grp * (4 / KG) + kk) * 2 + fhalf;
            const op16x8 a = *reinterpret_cast<const op16x8*>(cur + g_lds_off(wm * 32 + frow, ch));
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const op16x8 b = *reinterpret_cast<const op16x8*>(cur + 64 * 128 + g_lds_off(wn * 32 * NJ + j * 32 + frow, ch));
                acc[j] = CC_MFMA_32x32x16(a, b, acc[j]);
            }
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
#undef S64_ISSUE
    __syncthreads();
    // accumulator (row = 8 (r>>2) + 4 (lane>>5) + (r&3), col = lane&31) -> wave-private LDS strip -> 8 consecutive columns per lane
    constexpr int SLD = 32 * NJ + 4, CG = 4 * NJ, RPP = 64 / CG;
    if constexpr (KG == 2) {
        // group 1 hands over rows 0-15 (registers 0-7), group 0 rows 16-31 (registers 8-15): [reg][thread] float2 pairs
        float* xo = reinterpret_cast<float*>(sm64 + 32768) + grp * (8 * NJ * G_THREADS) + (tid & (G_THREADS - 1));
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 8; r++) xo[(j * 8 + r) * G_THREADS] = acc[j][grp ? r : 8 + r];
        __syncthreads();
        const float* xi = reinterpret_cast<const float*>(sm64 + 32768) + (grp ^ 1) * (8 * NJ * G_THREADS) + (tid & (G_THREADS - 1));
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float o = xi[(j * 8 + r) * G_THREADS];
                if (grp) acc[j][8 + r] += o; else acc[j][r] += o;
            }
    }
    float* strip = reinterpret_cast<float*>(sm64) + wave * (32 * SLD);
    constexpr int R0 = 0, RN = 16 / KG;                 // registers this wave finishes: [8 grp, 8 grp + RN) for KG = 2, all 16 otherwise
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int rr = R0; rr < RN; rr++) {
            const int r = (KG == 2 ? 8 * grp : 0) + rr;
            float val;
            if constexpr (KG == 2) val = grp ? acc[j][8 + rr] : acc[j][rr]; else val = acc[j][rr];
            strip[((r >> 2) * 8 + fhalf * 4 + (r & 3)) * SLD + j * 32 + frow] = val;
        }
#pragma unroll
    for (int ps = 0; ps < 32 / RPP / KG; ps++) {
        const int lr = (KG == 2 ? 16 * grp : 0) + ps * RPP + lane / CG, c8 = lane % CG;
        const float4 x = *reinterpret_cast<const float4*>(strip + lr * SLD + c8 * 8);
        const float4 y = *reinterpret_cast<const float4*>(strip + lr * SLD + c8 * 8 + 4);
        float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        epi(m0 + wm * 32 + lr, n0 + wn * 32 * NJ + c8 * 8, v);
    }
}

// 64 x 64 skinny kernel, K split over the waves: every wave computes the WHOLE tile for one 16-k quarter of each K-tile (2 + 2 fragment
// reads and 4 MFMAs per step and wave).  The 2 x 2 wave layout above reads 32 KiB of fragments per step through the CU's 128 B/clk
// LDS port — 322 of its ~600 cycles per step (tools/probes/step_timing.hip) — this one 16 KiB: 430 cycles per step.  The price is a
// cross-wave reduction of the four partial tiles through LDS at the end (~4 steps' worth), so gemm_nt_skinny picks it only for
// >= 12 K-tiles per block and grids of at most one block per CU (two co-resident blocks lose: the probe shows 730 vs 650 cycles).
template <class Epi>
__global__ __launch_bounds__(G_THREADS, 2) void gemm_nt_s64kw_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B, GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smkw[];
    constexpr int NS = 4, STAGE = 128 * 128, P = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (g.N + 63) / 64, tiles_m = (g.M + 63) / 64;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * 64, n0 = tn * 64;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#define KW_ISSUE(T, SLOT) glds_pair<64, 64, 4>(A, g.lda, g.M, m0, B, g.ldb, g.N, n0, kbeg + (T)*G_BK, smkw + (SLOT) * STAGE, wave, lane)
#pragma unroll
    for (int t = 0; t < NS - 1; t++)
        if (t < nk) KW_ISSUE(t, t);
    const int frow = lane & 31, ch = wave * 2 + (lane >> 5);
    int slot = 0, islot = NS - 1;
    for (int kt = 0; kt < nk; kt++) {
        const int rem = min(nk - 1, kt + NS - 2) - kt;
        if (rem >= 2) s_wait_vm<2 * P>();
        else if (rem == 1) s_wait_vm<P>();
        else s_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const char* cur = smkw + slot * STAGE;
        op16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            a[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(i * 32 + frow, ch));
            b[i] = *reinterpret_cast<const op16x8*>(cur + 64 * 128 + g_lds_off(i * 32 + frow, ch));
        }
        if (kt + NS - 1 < nk) KW_ISSUE(kt + NS - 1, islot);       // after the reads are issued: the DMA issue cycles hide their latency
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = CC_MFMA_32x32x16(a[i], b[j], acc[i][j]);
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
#undef KW_ISSUE
    __syncthreads();
    // partial tiles -> LDS as [wave][tile 2i+j][register][lane] floats (64 KiB = the four stages); element (row, col) of a tile sits in
    // register 4 (row>>3) + (row&3) of lane (col&31) + 32 ((row>>2)&1), so 8 consecutive columns are 8 consecutive floats
    float* part = reinterpret_cast<float*>(smkw);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) part[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    // wave w finishes rows [16 w, 16 w + 16) of the tile: lane -> (row, 8-column group), two passes
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
        const int row = wave * 16 + ps * 8 + (lane >> 3), c8 = lane & 7;
        const int rr = row & 31, reg = (rr >> 3) * 4 + (rr & 3), ls = (c8 & 3) * 8 + 32 * ((rr >> 2) & 1), tile = (row >> 5) * 2 + (c8 >> 2);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int pw = 0; pw < 4; pw++) {
            const float* src = part + ((pw * 4 + tile) * 16 + reg) * 64 + ls;
            const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
            v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
        }
        epi(m0 + row, n0 + c8 * 8, v);
    }
}

// 80 x 64 tiles, K split over the waves (round 5): the wave-quantisation fix for c_fc at M = 320.  With 64 x 64 tiles 320 x 4096 is 320 blocks
// on 256 CUs — 64 CUs carry two blocks and the launch takes as long as they do (12.7 us against 9.2 us for c_attn's 240 single blocks of the
// same depth); 320 rows as FOUR 80-row tiles are exactly 4 x 64 = 256 blocks, one per CU (hipBLASLt's pick for the shape, MT64x96, makes the
// same trade the other way round: 215 blocks; profiles/r05_j_*).  The LDS image of A is 96 rows (three 32-row MFMA fragments; the DMA clamps
// rows beyond M, rows 80-95 of a tile belong to the next tile and their products are dropped), a stage is 20 KiB, the cross-wave reduction
// of the 4 x 6 partial tiles needs 96 KiB.  Otherwise gemm_nt_s64kw_kernel: every wave computes the whole tile for one 16-k quarter of a K-tile.
template <class Epi>
__global__ __launch_bounds__(G_THREADS, 1) void gemm_nt_s80kw_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B, GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smk8[];
    constexpr int NS = 4, STAGE = (96 + 64) * 128, P = 5, BMT = 80;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (g.N + 63) / 64, tiles_m = (g.M + BMT - 1) / BMT;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * BMT, n0 = tn * 64;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    f32x16 acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
#define KW_ISSUE(T, SLOT) glds_pair<96, 64, 4>(A, g.lda, g.M, m0, B, g.ldb, g.N, n0, kbeg + (T)*G_BK, smk8 + (SLOT) * STAGE, wave, lane)
#pragma unroll
    for (int t = 0; t < NS - 1; t++)
        if (t < nk) KW_ISSUE(t, t);
    const int frow = lane & 31, ch = wave * 2 + (lane >> 5);
    int slot = 0, islot = NS - 1;
    for (int kt = 0; kt < nk; kt++) {
        const int rem = min(nk - 1, kt + NS - 2) - kt;
        if (rem >= 2) s_wait_vm<2 * P>();
        else if (rem == 1) s_wait_vm<P>();
        else s_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const char* cur = smk8 + slot * STAGE;
        op16x8 a[3], b[2];
#pragma unroll
        for (int i = 0; i < 3; i++) a[i] = *reinterpret_cast<const op16x8*>(cur + g_lds_off(i * 32 + frow, ch));
#pragma unroll
        for (int j = 0; j < 2; j++) b[j] = *reinterpret_cast<const op16x8*>(cur + 96 * 128 + g_lds_off(j * 32 + frow, ch));
        if (kt + NS - 1 < nk) KW_ISSUE(kt + NS - 1, islot);       // after the reads are issued: the DMA issue cycles hide their latency
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = CC_MFMA_32x32x16(a[i], b[j], acc[i][j]);
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
    }
#undef KW_ISSUE
    __syncthreads();
    // partial tiles -> LDS as [wave][tile 2i+j][register][lane] floats (96 KiB); element (row, col) of a 32 x 32 tile sits in register
    // 4 (row>>3) + (row&3) of lane (col&31) + 32 ((row>>2)&1), so 8 consecutive columns are 8 consecutive floats
    float* part = reinterpret_cast<float*>(smk8);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) part[((wave * 6 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    // 80 rows x 8 column groups of 8 = 640 items over 256 threads
#pragma unroll
    for (int ps = 0; ps < 3; ps++) {
        const int item = ps * G_THREADS + tid, row = item >> 3, c8 = item & 7;
        if (row < BMT) {
            const int rr = row & 31, reg = (rr >> 3) * 4 + (rr & 3), ls = (c8 & 3) * 8 + 32 * ((rr >> 2) & 1), tile = (row >> 5) * 2 + (c8 >> 2);
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int pw = 0; pw < 4; pw++) {
                const float* src = part + ((pw * 6 + tile) * 16 + reg) * 64 + ls;
                const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
                v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
            }
            epi(m0 + row, n0 + c8 * 8, v);
        }
    }
}

// The same 64 x 64 / K-over-the-waves kernel with the WEIGHT operand taken out of LDS (round 5): B comes from a
// fragment-ordered image (k_skinny_image, cc_decode_image) — per (64-column tile, 64-k tile, wave = 16-k quarter, 32-column half) one 1-KiB
// piece in which lane l holds W[n0 + 32 i + (l & 31)][k0 + 16 w + 8 (l >> 5) .. + 7], i.e. exactly the lane's v_mfma_f32_32x32x16 operand —
// and is loaded global -> VGPR, fully coalesced, three K-tiles ahead.  No DMA, no LDS write, no ds_read for B: the LDS carries the 8-KiB
// activation stage only (half the DMA bytes and half the fragment reads of gemm_nt_s64kw_kernel, whose K-step the fragment reads bound).
// The B loads are inline asm beside the LDS-DMA of A (hipcc drains the DMA queue at any ordinary VGPR load it sees next to a
// global_load_lds), so both queues are counted by hand: per K-tile a wave issues 2 DMA + 2 B loads, in order; tile t is complete when at
// most 4 * (tiles issued after t) operations are outstanding.  The B registers are a ring of 4 tiles with static indices (loop unrolled x 4).
// In the decode chain the weights are COLD (708 MB per generated position stream from HBM once; the hot-cache micro-benchmark hides it): a
// block's K loop is then a chain of HBM round trips, nk / (tiles in flight) of them — 16 tiles at 3 in flight = 5-6 round trips of ~2 us.
// With B in registers the prefetch depth costs registers only, not LDS: a ring of RB K-tiles (2 KiB per wave and tile).  For the depth to
// be real the B loads must not share a vmcnt queue with the activation DMA (loads return in order: waiting for the A stage of tile t would
// also wait for every B tile requested before it), so a FIFTH wave issues all of the A DMA (and owns its waits); waves 0-3 issue nothing
// but their B fragments and count them exactly: vmcnt(2 (RB - 1)).
typedef unsigned kb_u32x4 __attribute__((ext_vector_type(4)));
#ifdef CC_EXPERIMENTS   // lab build only (round 5: bit-identical, 1-3 % slower in the real decode chain — HISTORY.md)
template <class Epi, int RB>
__global__ __launch_bounds__(320, RB == 16 ? 2 : 3) void gemm_nt_s64kwb_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ Bimg, GemmShape g, Epi epi) {
    static_assert(RB == 4 || RB == 8 || RB == 16, "ring depth");
    extern __shared__ __attribute__((aligned(1024))) char smkb[];
    constexpr int STAGE = 64 * 128;                       // one activation stage: 64 rows x 128 B; four of them, then reused by the epilogue (64 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (g.N + 63) / 64, tiles_m = (g.M + 63) / 64;
    int tm, tn;
    tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * 64, n0 = tn * 64;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    const int ktot = g.K / G_BK;
    const int last = nk - 1;
    const int ngrp = (nk + RB - 1) / RB;                  // whole groups of RB steps; the steps beyond the last K-tile only keep the counts
    if (wave == 4) {
        // ---- loader wave: the activation stages (8 DMA instructions each), three K-tiles ahead
#define KB_A(T, U)                                                                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; i_++) {                                                                                \
        const int row_ = i_ * 8 + (lane >> 3);                                                                                        \
        const int chunk_ = (lane & 7) ^ ((row_ >> 1) & 7) ^ ((row_ >> 4) & 3);                                                        \
        const op16_t* src_ = A + (size_t)min(m0 + row_, g.M - 1) * g.lda + kbeg + (T)*G_BK + chunk_ * 8;                              \
        __builtin_amdgcn_global_load_lds((gptr_t)src_, (lptr_t)(smkb + (U)*STAGE + i_ * 1024), 16, 0, 0);                             \
    }
        KB_A(0, 0)
        KB_A(min(1, last), 1)
        KB_A(min(2, last), 2)
        for (int t = 0; t < ngrp * RB; t++) {
            s_wait_vm<16>();                               // tile t has landed (t + 1, t + 2 may be in flight)
            __builtin_amdgcn_s_barrier();
            KB_A(min(t + 3, last), (t + 3) & 3)            // that stage held K-tile t - 1: every wave is past it
        }
#undef KB_A
        s_wait_vm<0>();
    } else {
        // this wave's two fragments of K-tile t: Bimg + (((tn * ktot + kbeg / 64 + t) * 4 + wave) * 2 + i) * 512 elements
        const op16_t* bptr = Bimg + ((size_t)(tn * ktot + kbeg / G_BK) * 4 + wave) * 1024 + lane * 8;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        // The ring registers must never pass through a copy: an asm load is asynchronous behind the compiler's back, so a v_mov of a ring
        // register (a PHI at a branch that issues conditionally, or at the exit into a separate tail block) reads it before the data lands
        // and frees it for other values that the landing load then overwrites (measured: a fault on the first shape whose K-tile count was
        // not a multiple of the unroll).  So: ONE loop of whole groups, every step issues its two loads (K-tile index clamped), the wait is a
        // constant, and only the fragment reads + MFMAs of a step beyond the last K-tile are skipped (a branch over the accumulators only).
        kb_u32x4 br[RB][2];
#define KB_B(T, U)                                                                                                                    \
    {                                                                                                                                 \
        const op16_t* bp_ = bptr + (size_t)(T)*4096;                                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(br[U][0]) : "v"(bp_) : "memory");                                       \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(br[U][1]) : "v"(bp_) : "memory");                           \
    }
#pragma unroll
        for (int t = 0; t < RB; t++) KB_B(min(t, last), t)
        const int frow = lane & 31, ch = wave * 2 + (lane >> 5);
        for (int kt0 = 0; kt0 < ngrp * RB; kt0 += RB) {
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int kt = kt0 + u;
                if constexpr (RB == 4) asm volatile("s_waitcnt vmcnt(6)" : "+v"(br[u][0]), "+v"(br[u][1])::"memory");
                else if constexpr (RB == 8) asm volatile("s_waitcnt vmcnt(14)" : "+v"(br[u][0]), "+v"(br[u][1])::"memory");
                else asm volatile("s_waitcnt vmcnt(30)" : "+v"(br[u][0]), "+v"(br[u][1])::"memory");
                __builtin_amdgcn_s_barrier();
                if (kt <= last) {
                    const char* cur_ = smkb + (kt & 3) * STAGE;
                    const op16x8 a0_ = *reinterpret_cast<const op16x8*>(cur_ + g_lds_off(frow, ch));
                    const op16x8 a1_ = *reinterpret_cast<const op16x8*>(cur_ + g_lds_off(32 + frow, ch));
                    const op16x8 b0_ = __builtin_bit_cast(op16x8, br[u][0]), b1_ = __builtin_bit_cast(op16x8, br[u][1]);
                    acc[0][0] = CC_MFMA_32x32x16(a0_, b0_, acc[0][0]);
                    acc[0][1] = CC_MFMA_32x32x16(a0_, b1_, acc[0][1]);
                    acc[1][0] = CC_MFMA_32x32x16(a1_, b0_, acc[1][0]);
                    acc[1][1] = CC_MFMA_32x32x16(a1_, b1_, acc[1][1]);
                }
                KB_B(min(kt + RB, last), u)                 // the slot just consumed
            }
        }
#undef KB_B
        // the (redundant) loads still in flight land in the ring registers: they stay the ring's until this wait
#pragma unroll
        for (int u = 0; u < RB; u++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(br[u][0]), "+v"(br[u][1])::"memory");
        // partial tiles -> LDS (after every wave has left the K loop: the stages are dead)
        __builtin_amdgcn_s_barrier();
        float* part = reinterpret_cast<float*>(smkb);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) part[((wave * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    if (wave == 4) __builtin_amdgcn_s_barrier();           // (the loader's share of the barrier above)
    __syncthreads();
    if (wave < 4) {
        const float* part = reinterpret_cast<const float*>(smkb);
#pragma unroll
        for (int ps = 0; ps < 2; ps++) {
            const int row = wave * 16 + ps * 8 + (lane >> 3), c8 = lane & 7;
            const int rr = row & 31, reg = (rr >> 3) * 4 + (rr & 3), ls = (c8 & 3) * 8 + 32 * ((rr >> 2) & 1), tile = (row >> 5) * 2 + (c8 >> 2);
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int pw = 0; pw < 4; pw++) {
                const float* src = part + ((pw * 4 + tile) * 16 + reg) * 64 + ls;
                const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
                v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
            }
            epi(m0 + row, n0 + c8 * 8, v);
        }
    }
}
// the image a 64-column x 64-k tile of W [N][K] takes in cc_decode_image: element offset of (tile column tn, K-tile kt, wave w, half i, lane l)
// = (((tn * (K / 64) + kt) * 4 + w) * 2 + i) * 512 + l * 8 — a permutation of the matrix, so the image has the matrix's size
static __global__ __launch_bounds__(256) void k_skinny_image(const op16_t* __restrict__ W, op16_t* __restrict__ img, int N, int K) {
    const size_t total = (size_t)N * K / 8;
    const int ktot = K / 64;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int l = (int)(e & 63);
        size_t f = e >> 6;
        const int i = (int)(f & 1); f >>= 1;
        const int w = (int)(f & 3); f >>= 2;
        const int kt = (int)(f % ktot), tn = (int)(f / ktot);
        const op16_t* src = W + (size_t)(tn * 64 + i * 32 + (l & 31)) * K + kt * 64 + w * 16 + (l >> 5) * 8;
        reinterpret_cast<kb_u32x4*>(img)[e] = *reinterpret_cast<const kb_u32x4*>(src);
    }
}

#endif  // CC_EXPERIMENTS

// ------------------------------------------------------------------------------------------------
// 256-row kernels for large outputs: 8 waves, wave tile 128 x 16 NJ (8 x NJ MFMA tiles; block tile 256 x 256 for NJ = 4,
// 256 x 192 for NJ = 3), K-tile 32, FOUR LDS stages of 32 KiB.  Per FLOP the 256 x 256 form moves half the L2->LDS bytes and
// 3/4 of the fragment reads of the 128 x 128 kernel — the two resources the ablation shows saturated there.  One block per CU
// means no second block to hide a barrier stall, so the two wave groups (rows 0-127 / 128-255; one wave of each per SIMD) are
// locked one segment apart by the block barrier; a segment is either the 8 + NJ fragment reads of a K-tile or its 8 NJ MFMAs:
//       segment 2t:   g0 reads(t)  | g1 MFMA(t-1)          segment 2t+1:   g0 MFMA(t) | g1 reads(t)
// DMA (global_load_lds) of tile t+3 is issued by the group that is in its READ segment — group 0 streams the A tile in segment
// 2t, group 1 the B tile in segment 2t+1; the buffer's last reader finished in segment 2t-1 — never in front of a group's MFMAs
// (4 DMA instructions cost ~300 issue cycles: 1145 -> 1230 TFLOP/s at 8192^3).  Tile t+1 is awaited with a COUNTED vmcnt before
// the barrier ending segment 2t+1, so two tiles stay in flight across the barriers (raw s_barrier; a __syncthreads() would
// drain the DMA queue).  The two groups run separate copies of the loop (one loop with per-segment role branches made hipcc
// spill 90 registers).  TT = true is the weight-gradient form (both operands K-strided), see H_ISSUE / h_tt_read.
// NT LDS image [row][32 k] (64-B rows) with the bank-group-exact XOR key h_swz below.
// ------------------------------------------------------------------------------------------------
#ifndef CC_STAG_READ_FIRST
#define CC_STAG_READ_FIRST 1   // fragment reads of tile t are issued before the DMA of tile t+3: the DMA issue stalls (queue back-pressure) then cover the read latency; +1-2 % on the K <= 4096 shapes, A/B in HISTORY.md 4.5
#endif
constexpr int H_BM = 256, H_BN = 256, H_BK = 32, H_NS = 4, H_STAGE = (H_BM + H_BN) * H_BK * 2;   // 32 KiB per stage, 128 KiB total (5 stages = all 160 KiB measured 2-3 % slower)
// The B tile may be narrower: NJ MFMA column tiles per wave -> block tile 256 x (64 NJ); NJ = 3 gives 256 x 192 for N = 768-like widths.
// 64-B rows, 4 chunks of 16 B.  ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (guide, LDS
// table): with lane = (row & 15) + 16 * chunk a group holds rows 0-3 and 12-15 at chunk c and rows 4-11 at chunk c^1, so the XOR
// key has to be 0,0,3,3 for row>>2 = 0..3 (not 0,1,2,3) for the 16 lanes to land on 16 different 16-B bank groups.
__device__ __forceinline__ int h_swz(int row) { return (row & 8) ? 3 : 0; }
__device__ __forceinline__ int h_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ h_swz(row)) << 4); }
// DMA issue is done by the group that is in its fragment-read segment (never ahead of a group's MFMAs: 4 DMA instructions cost
// ~600 issue cycles): group 0 streams the A tile, group 1 the B tile; wave w of a group takes 1-KiB segments w, w+4, w+8, w+12.
// TT (both operands K-strided: A is [K][M], B is [K][N] — the weight-gradient GEMM): the LDS image of a stage is [32 k][256 cols]
// (512-B rows), filled by the same 16 DMA instructions per operand (2 k-rows each), and the fragments are read with
// ds_read_b64_tr_b16: a 16-lane group fetches a [4 k][16 col] block, 8 B per lane at row k0 + (lane>>2), column 4 (lane&3), and
// lane j receives column j's four k values — the MFMA operand layout without any register transpose (probed on MI355X).
// A 32-lane half of that read touches 8 rows x 32 B; rows are 512 B apart, so the 32-B slot index inside each 256-B window is
// XORed with g(k) = (k & 3) | ((k >> 3 & 1) << 2), distinct for the 8 rows of a half: conflict-free.
__device__ __forceinline__ int h_tt_g(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
#define H_ISSUE(T, IS_A)                                                                                                 \
    {                                                                                                                    \
        char* st_ = smem + ((T) % H_NS) * SSTAGE + ((IS_A) ? 0 : SBM * H_BK * 2);                                      \
        int k0_ = kbeg + (T)*H_BK;                                                                                       \
        if constexpr (X3F) k0_ = (x3c0_ + ((T) >> 1)) * H_BK + (((T) & 1) ? ((IS_A) ? 2 * x3k_ : x3k_) : 0);   /* hi stage, then lo stage of the same K chunk */ \
        _Pragma("unroll") for (int i_ = 0; i_ < ((IS_A) ? (NI + 1) / 2 : NJ); i_++) {                                        \
            const int seg = (IS_A) ? min(wn + 4 * i_, 2 * NI - 1) : wn + 4 * i_;   /* 2 NI (A) or 4 NJ (B) segments of 1 KiB; odd NI: two waves re-issue the last one (same bytes, same place) so that every wave counts the same DMA instructions */ \
            const op16_t* src;                                                                                           \
            if constexpr (TT) {                                                                                          \
                const int kr = seg * 2 + (lane >> 5);           /* 2 k-rows of 512 B per segment */                       \
                const int c = (lane & 31) ^ (h_tt_g(kr) << 1);                                                            \
                src = (IS_A) ? A + (size_t)(k0_ + kr) * g.lda + min(m0 + c * 8, g.M - 8)                                  \
                             : B + (size_t)(k0_ + kr) * g.ldb + min(n0 + c * 8, g.N - 8);                                 \
            } else {                                                                                                     \
                const int row = seg * 16 + (lane >> 2);         /* 16 rows x 64 B per segment */                          \
                const int chunk = (lane & 3) ^ h_swz(row);                                                               \
                src = (IS_A) ? A + (size_t)min(m0 + row, g.M - 1) * g.lda + k0_ + chunk * 8                               \
                             : B + (size_t)min(n0 + row, g.N - 1) * g.ldb + k0_ + chunk * 8;                              \
            }                                                                                                            \
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st_ + seg * 1024), 16, 0, 0);                         \
        }                                                                                                                \
    }
// The transpose reads are issued as inline asm: through the builtin the compiler cannot prove they do not alias the pending
// global_load_lds writes and puts an s_waitcnt vmcnt(0) in front of every fragment-read segment (measured: 468 instead of ~900
// TFLOP/s).  The asm results are only combined into MFMA operands (H_TT_OPER) after the segment's own s_waitcnt lgkmcnt(0).
typedef __attribute__((ext_vector_type(2))) int h_i32x2;
struct HTTFrag { h_i32x2 lo, hi; };          // k = 8 kg + 0..3 and 8 kg + 4..7 of one column
__device__ __forceinline__ HTTFrag h_tt_read(const char* tile, int lane_base, int xoff) {
    typedef __attribute__((address_space(3))) const char* lp_t;
    const unsigned a = (unsigned)(size_t)(lp_t)(tile + lane_base + xoff);
    HTTFrag f;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(f.hi) : "v"(a) : "memory");
    return f;
}
__device__ __forceinline__ op16x8 h_tt_oper(const HTTFrag& f) {
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    const i32x4_t v = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(op16x8, v);
}
#define H_LOADF(T)                                                                                                       \
    {                                                                                                                    \
        const char* ca_ = smem + ((T) % H_NS) * SSTAGE;                                                                 \
        const char* cb_ = ca_ + SBM * H_BK * 2;                                                                         \
        if constexpr (TT) {                                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) taf[i_] = h_tt_read(ca_, tt_base, ((grp * 16 + 2 * i_) ^ tt_gx) << 4);  \
            _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++) tbf[j_] = h_tt_read(cb_, tt_base, ((wn * 8 + 2 * j_) ^ tt_gx) << 4);  \
        } else {                                                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) af[i_] = *reinterpret_cast<const op16x8*>(ca_ + h_lds_off(arow + i_ * 16 + frow, fchunk)); \
            _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++) bfr[j_] = *reinterpret_cast<const op16x8*>(cb_ + h_lds_off(bcol + j_ * 16 + frow, fchunk)); \
        }                                                                                                                \
    }
#define H_MFMA(T)                                                                                                        \
    {                                                                                                      \
        if constexpr (TT) {                                                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) af[i_] = h_tt_oper(taf[i_]);                                 \
            _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++) bfr[j_] = h_tt_oper(tbf[j_]);                               \
        }                                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++)                \
            acc[i_][j_] = CC_MFMA_16x16x32(bfr[j_], af[i_], acc[i_][j_]);                \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
    }
// X3F: the lo stage of a K chunk — its fragments go to their own registers, its MFMA segment is A_hi B_lo + A_lo B_hi
#define H_LOADF_LO(T)                                                                                                    \
    {                                                                                                                    \
        const char* ca_ = smem + ((T) % H_NS) * SSTAGE;                                                                 \
        const char* cb_ = ca_ + SBM * H_BK * 2;                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) af2[X3F ? i_ : 0] = *reinterpret_cast<const op16x8*>(ca_ + h_lds_off(arow + i_ * 16 + frow, fchunk)); \
        _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++) bf2[X3F ? j_ : 0] = *reinterpret_cast<const op16x8*>(cb_ + h_lds_off(bcol + j_ * 16 + frow, fchunk)); \
    }
#define H_MFMA_LO()                                                                                                      \
    {                                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++)                \
            acc[i_][j_] = CC_MFMA_16x16x32(bf2[X3F ? j_ : 0], af[i_], acc[i_][j_]);                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < NI; i_++) _Pragma("unroll") for (int j_ = 0; j_ < NJ; j_++)                \
            acc[i_][j_] = CC_MFMA_16x16x32(bfr[j_], af2[X3F ? i_ : 0], acc[i_][j_]);                                       \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
    }
#define H_SEGEND()                                                                                                       \
    {                                                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }
// tile t+1 must have landed; tiles t+2 .. t+H_NS-1 (CNT DMA instructions per wave each: NI / 2 in group 0, NJ in group 1) may stay in flight
#define H_WAIT(T, CNT)                                                                                                   \
    {                                                                                                                    \
        const int rem_ = min(nk - 1, (T) + H_NS - 1) - ((T) + 1);                                                        \
        if (rem_ >= 3) s_wait_vm<3 * (CNT)>();                                                                           \
        else if (rem_ == 2) s_wait_vm<2 * (CNT)>();                                                                      \
        else if (rem_ == 1) s_wait_vm<(CNT)>();                                                                          \
        else s_wait_vm<0>();                                                                                             \
    }
// tools/probes/stag256_timing.hip compiles this header with CC_STAMP: cycle counts per phase of the main loop, wave 0 of either group
#ifdef CC_STAMP
__device__ unsigned long long cc_stamp_buf[2 * 8];
#define H_STAMP_DECL unsigned long long st_last_ = __builtin_readcyclecounter(), st_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define H_STAMP(I) { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc_[I] += n_ - st_last_; st_last_ = n_; }
#define H_STAMP_OUT if (blockIdx.x == 7 && wn == 0 && lane == 0) { for (int i_ = 0; i_ < 8; i_++) cc_stamp_buf[grp * 8 + i_] = st_acc_[i_]; }
#else
#define H_STAMP_DECL
#define H_STAMP(I)
#define H_STAMP_OUT
#endif
// body shared by the single-problem kernels and the grouped weight-gradient kernel: `tile` = logical tile of this block inside its
// problem (XCD remap applied by the caller), `zslice` = its K slice
// X3F (bf16x3 build, 256 x 192 form, one K slice): the operands are the [hi | hi | lo] / [hi | lo | hi] images over K' = 3 K.  Instead of walking
// K' as three passes (hi hi, hi lo, lo hi: three stages of DMA and fragment reads per K chunk), a chunk is TWO stages — its hi tiles, then
// its lo tiles — and the lo stage's MFMA segment runs A_hi B_lo + A_lo B_hi with the hi fragments kept in registers: 2/3 of the DMA bytes,
// fragment reads and segment hand-offs for the same 3 MFMA products.  (44 more registers: fits the 178-register 256 x 192 form only.)
template <class Epi, int NJ, bool TT, int NI = 8, bool X3F = false>
__device__ __forceinline__ void gemm_stag256_body(const op16_t* __restrict__ A, const op16_t* __restrict__ B, const GemmShape& g, const Epi& epi,
                                                  int tile, int zslice, char* smem) {
    static_assert(NJ >= 2 && NJ <= 4 && (NI == 8 || NI == 10 || (NI == 5 && NJ == 4)), "wave tile is (16 NI) x (16 NJ)");
    static_assert(!TT || (NJ == 4 && NI == 8), "the K-strided image is laid out for 256 x 256 tiles");
    static_assert(!X3F || (!TT && NJ == 3 && NI == 8), "the fused split-bf16 form is the 256 x 192 NT kernel's");
    constexpr int BN = 64 * NJ, SBM = 32 * NI, SSTAGE = (SBM + H_BN) * H_BK * 2;      // NI = 10: 320-row tiles, 36 KiB stages
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int arow = grp * (16 * NI), bcol = wn * (16 * NJ);
    const int tiles_n = (g.N + BN - 1) / BN, tiles_m = (g.M + SBM - 1) / SBM;
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * SBM, n0 = tn * BN;
    const int kbeg = zslice * g.k_chunk;                      // split-K slice (k_chunk is a multiple of 64)
    const int x3k_ = g.K / 3;                                 // X3F: the logical K (g.K = K' = 3 K); K slices are ranges of its 32-wide chunks
    const int x3nc_ = x3k_ / H_BK, x3per_ = (x3nc_ + (int)gridDim.z - 1) / (int)gridDim.z;
    const int x3c0_ = zslice * x3per_, x3c1_ = min(x3nc_, x3c0_ + x3per_);
    (void)x3c0_;
    const int nk = X3F ? 2 * max(0, x3c1_ - x3c0_) : (min(g.K, kbeg + g.k_chunk) - kbeg) / H_BK;
    f32x4 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Functors with an additive fp32 input (the residual stream) can start the accumulators FROM it: the 39 MB of residual reads of a
    // 12800 x 768 launch then travel while the first operand tiles are in flight, instead of joining the store burst at the end of a
    // single-round launch.  Issued before the first DMA so that every later counted vmcnt wait implies these loads have landed.
    // (256 x 192 form only: the 256-wide forms have no registers to spare for the address arithmetic — 276 / 564 B of scratch measured)
    constexpr bool kAccInit = epi_acc_init<Epi>::value && ((NJ == 3 && NI == 8) || NI == 5) && !X3F;      // (the fused split-bf16 form has no registers to spare either: 54 spilled)
    if constexpr (kAccInit) {
        if (zslice == 0 && epi.init_from_input()) {
            const int q_ = lane >> 4, rl_ = lane & 15;
#pragma unroll
            for (int i = 0; i < NI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) epi.init4(m0 + arow + i * 16 + rl_, n0 + bcol + j * 16 + q_ * 4, acc[i][j]);
        }
    }
    op16x8 af[NI], bfr[NJ];
    op16x8 af2[X3F ? NI : 1], bf2[X3F ? NJ : 1];              // X3F: the lo stage's fragments
    (void)af2; (void)bf2;
    HTTFrag taf[TT ? NI : 1], tbf[TT ? NJ : 1];
    (void)taf; (void)tbf;
    const int frow = lane & 15, fchunk = lane >> 4;
    // TT fragment addressing: k row 8 fchunk + (frow >> 2), byte 16 ((frow & 3) >> 1) + 8 (frow & 1) inside the 32-B slot
    const int tt_base = (8 * fchunk + (frow >> 2)) * 512 + ((frow & 3) >> 1) * 16 + (frow & 1) * 8;
    const int tt_gx = h_tt_g(8 * fchunk + (frow >> 2)) << 1;
    (void)tt_base; (void)tt_gx;
    H_STAMP_DECL
    if constexpr (X3F) {
        // two stages per K chunk, parity known at compile time (a run-time parity test in one loop body kept both fragment sets and their
        // copies live: 171-285 spilled registers)
#define H_G0_STEP(T, LOADF, MFMA)                                                                                        \
        {                                                                                                                \
            LOADF(T);                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            if ((T) + H_NS - 1 < nk) H_ISSUE((T) + H_NS - 1, true);                                                      \
            H_SEGEND();                                                                                                  \
            MFMA;                                                                                                        \
            H_WAIT(T, (NI + 1) / 2);                                                                                           \
            H_SEGEND();                                                                                                  \
        }
#define H_G1_STEP(T, LOADF, MFMA_PREV)                                                                                   \
        {                                                                                                                \
            MFMA_PREV;                                                                                                   \
            H_SEGEND();                                                                                                  \
            LOADF(T);                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            if ((T) + H_NS - 1 < nk) H_ISSUE((T) + H_NS - 1, false);                                                     \
            H_WAIT(T, NJ);                                                                                               \
            H_SEGEND();                                                                                                  \
        }
        if (grp == 0) {
            if (nk > 0) H_ISSUE(0, true);                    // (an empty K slice writes a zero slab)
            if (nk > 1) H_ISSUE(1, true);
            if (nk > 2) H_ISSUE(2, true);
            H_WAIT(-1, (NI + 1) / 2);
            H_SEGEND();
            for (int t = 0; t < nk; t += 2) {
                H_G0_STEP(t, H_LOADF, H_MFMA(t))
                H_G0_STEP(t + 1, H_LOADF_LO, H_MFMA_LO())
            }
            H_SEGEND();
        } else {
            if (nk > 0) H_ISSUE(0, false);
            if (nk > 1) H_ISSUE(1, false);
            if (nk > 2) H_ISSUE(2, false);
            H_WAIT(-1, NJ);
            H_SEGEND();
            for (int t = 0; t < nk; t += 2) {
                H_G1_STEP(t, H_LOADF, if (t >= 1) H_MFMA_LO())
                H_G1_STEP(t + 1, H_LOADF_LO, H_MFMA(t))
            }
            if (nk > 0) H_MFMA_LO();
            H_SEGEND();
        }
#undef H_G0_STEP
#undef H_G1_STEP
    } else
    if (grp == 0) {
        H_ISSUE(0, true);
        if (nk > 1) H_ISSUE(1, true);
        if (nk > 2) H_ISSUE(2, true);
        if (H_NS > 4 && nk > 3) H_ISSUE(3, true);
        H_WAIT(-1, (NI + 1) / 2);
        H_SEGEND();
        H_STAMP(0)                                     // prologue
        for (int t = 0; t < nk; t++) {
#if CC_STAG_READ_FIRST
            H_LOADF(t);
            __builtin_amdgcn_sched_barrier(0);
            if (t + H_NS - 1 < nk) H_ISSUE(t + H_NS - 1, true);
            H_STAMP(1)
#else
            if (t + H_NS - 1 < nk) H_ISSUE(t + H_NS - 1, true);      // that buffer held tile t-1, last read (by group 1) in segment 2t-1
            H_STAMP(1)                                 // DMA issue
            H_LOADF(t);
#endif
#ifdef CC_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            H_STAMP(2)                                 // fragment reads + wait
            H_SEGEND();
            H_STAMP(3)                                 // barrier (end of read segment)
            H_MFMA(t);
            H_STAMP(4)                                 // MFMA issue
            H_WAIT(t, (NI + 1) / 2);
            H_STAMP(5)                                 // vmcnt wait
            H_SEGEND();
            H_STAMP(6)                                 // barrier (end of MFMA segment)
        }
        H_SEGEND();
    } else {
        H_ISSUE(0, false);
        if (nk > 1) H_ISSUE(1, false);
        if (nk > 2) H_ISSUE(2, false);
        if (H_NS > 4 && nk > 3) H_ISSUE(3, false);
        H_WAIT(-1, NJ);
        H_SEGEND();
        H_STAMP(0)
        for (int t = 0; t < nk; t++) {
            if (t >= 1) H_MFMA(t - 1);
            H_STAMP(4)
            H_SEGEND();
            H_STAMP(6)
#if CC_STAG_READ_FIRST
            H_LOADF(t);
            __builtin_amdgcn_sched_barrier(0);
            if (t + H_NS - 1 < nk) H_ISSUE(t + H_NS - 1, false);
            H_STAMP(1)
#else
            if (t + H_NS - 1 < nk) H_ISSUE(t + H_NS - 1, false);     // one segment after group 0's half of the same tile
            H_STAMP(1)
            H_LOADF(t);
#endif
#ifdef CC_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            H_STAMP(2)
            H_WAIT(t, NJ);
            H_STAMP(5)
            H_SEGEND();
            H_STAMP(3)
        }
        H_MFMA(nk - 1); H_SEGEND();
    }
    H_STAMP(0)
    if constexpr (epi_acc_init<Epi>::value && !kAccInit) {      // forms that did not start from the input: the epilogue adds it as before
        Epi e2 = epi;
        e2.acc_init = false;
        gemm_epilogue_regs<Epi, NI, NJ>(acc, lane, m0 + arow, n0 + bcol, e2);
    } else {
        gemm_epilogue_regs<Epi, NI, NJ>(acc, lane, m0 + arow, n0 + bcol, epi);
    }
    H_STAMP(7)                                         // epilogue
    H_STAMP_OUT
}
template <class Epi, int NJ, bool TT = false, int NI = 8, bool X3F = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_stag256_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B, GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    gemm_stag256_body<Epi, NJ, TT, NI, X3F>(A, B, g, epi, xcd_remap(blockIdx.x, gridDim.x), (int)blockIdx.z, smem);
}
#undef H_STAMP_DECL
#undef H_STAMP
#undef H_STAMP_OUT
#undef H_ISSUE
#undef H_LOADF
#undef H_MFMA
#undef H_LOADF_LO
#undef H_MFMA_LO
#undef H_SEGEND
#undef H_WAIT

#include "gemm_q4.hip.h"

// ------------------------------------------------------------------------------------------------
// TT 128 x 128 kernel (both operands K-strided, weight gradients below the 256-row kernel's break-even): the 4-stage small-grid
// pipeline above with the K-strided staging and transpose reads of the 256-row TT kernel.  LDS stage = A [64 k][128 cols] |
// B [64 k][128 cols] (256-B rows, 16 KiB each); one DMA instruction = 4 k-rows; the 32-B slot XOR g(k) covers the whole row.
// Fragment reads are software-pipelined one 32-deep K-step ahead (a wave is alone on its SIMD: nothing else hides the LDS latency).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ HTTFrag g_tt_read(const char* tile, int lane_base, int xoff) {
    typedef __attribute__((address_space(3))) const char* lp_t;
    const unsigned a = (unsigned)(size_t)(lp_t)(tile + lane_base + xoff);
    HTTFrag f;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f.hi) : "v"(a) : "memory");      // + 4 k-rows of 256 B
    return f;
}
__device__ __forceinline__ void glds_tile_tt(const op16_t* __restrict__ base, int ld, int cols, int c0, int k0, char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int seg = wave + 4 * i;                      // 16 segments of 1 KiB = 4 k-rows x 256 B
        const int kr = seg * 4 + (lane >> 4);
        const int c = (lane & 15) ^ (h_tt_g(kr) << 1);
        const op16_t* src = base + (size_t)(k0 + kr) * ld + min(c0 + c * 8, cols - 8);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + seg * 1024), 16, 0, 0);
    }
}
// body shared by the single-problem kernel and the grouped kernel: `tile` = logical tile index of this block inside its problem (XCD
// remap already applied by the caller), `zslice` = its K slice
template <class Epi>
__device__ __forceinline__ void gemm_tt_glds4_body(const op16_t* __restrict__ A, const op16_t* __restrict__ B, const GemmShape& g, const Epi& epi,
                                                   int tile, int zslice, char* smemt) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + G_BN - 1) / G_BN, tiles_m = (g.M + G_BM - 1) / G_BM;
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * G_BM, n0 = tn * G_BN;
    const int kbeg = zslice * g.k_chunk;
    const int nk = (min(g.K, kbeg + g.k_chunk) - kbeg) / G_BK;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define GT_ISSUE(T)                                                                                                      \
    {                                                                                                                    \
        char* st_ = smemt + ((T) & 3) * 2 * G_TILE_BYTES;                                                                \
        glds_tile_tt(A, g.lda, g.M, m0, kbeg + (T)*G_BK, st_, wave, lane);                                               \
        glds_tile_tt(B, g.ldb, g.N, n0, kbeg + (T)*G_BK, st_ + G_TILE_BYTES, wave, lane);                                \
    }
    GT_ISSUE(0);
    if (nk > 1) GT_ISSUE(1);
    if (nk > 2) GT_ISSUE(2);
    const int frow = lane & 15, fchunk = lane >> 4;
    const int tt_base = (8 * fchunk + (frow >> 2)) * 256 + ((frow & 3) >> 1) * 16 + (frow & 1) * 8;
    const int tt_gx = h_tt_g(8 * fchunk + (frow >> 2)) << 1;
    HTTFrag fa[2][4], fb[2][4];
#define GT_READ(SET, CUR, KS)                                                                                            \
    {                                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) fa[SET][i_] = g_tt_read((CUR) + (KS)*32 * 256, tt_base, ((wm * 8 + 2 * i_) ^ tt_gx) << 4);                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) fb[SET][j_] = g_tt_read((CUR) + G_TILE_BYTES + (KS)*32 * 256, tt_base, ((wn * 8 + 2 * j_) ^ tt_gx) << 4);  \
    }
#define GT_MFMA(SET)                                                                                                     \
    {                                                                                                                    \
        op16x8 af[4], bfr[4];                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) af[i_] = h_tt_oper(fa[SET][i_]);                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) bfr[j_] = h_tt_oper(fb[SET][j_]);                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++)                 \
            acc[i_][j_] = CC_MFMA_16x16x32(af[i_], bfr[j_], acc[i_][j_]);                \
    }
#define GT_LGKM0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    for (int kt = 0; kt < nk; kt++) {
        const int rem = min(nk - 1, kt + 2) - kt;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 3 < nk) GT_ISSUE(kt + 3);
        const char* cur = smemt + (kt & 3) * 2 * G_TILE_BYTES;
        GT_READ(0, cur, 0);
        GT_LGKM0();
        GT_READ(1, cur, 1);                    // K-step 1 fragments in flight under K-step 0's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        GT_MFMA(0);
        GT_LGKM0();
        GT_MFMA(1);
    }
#undef GT_ISSUE
#undef GT_READ
#undef GT_MFMA
#undef GT_LGKM0
    __syncthreads();
    gemm_epilogue(acc, smemt, wave, lane, m0 + wm * 64, n0 + wn * 64, epi);
}
template <class Epi>
__global__ __launch_bounds__(G_THREADS, 1) void gemm_tt_glds4_kernel(const op16_t* __restrict__ A, const op16_t* __restrict__ B,
                                                                      GemmShape g, Epi epi) {
    extern __shared__ __attribute__((aligned(1024))) char smemt[];
    gemm_tt_glds4_body(A, B, g, epi, xcd_remap(blockIdx.x, gridDim.x), (int)blockIdx.z, smemt);
}


// ------------------------------------------------------------------------------------------------
// Epilogues.  operator()(row, col, v[8]) is called by EVERY lane (wave-uniform call site): lanes q..q+7 of
// a wave hold the 64 consecutive columns [col&~63, +64) of one row, so row-wise reductions are 3 shuffles.
// ------------------------------------------------------------------------------------------------

// C(bf16) = act(acc + bias); optionally also stores the pre-activation (needed by gelu backward).
__device__ __forceinline__ void load_bias8(const float* bias, int col, int Ns, float (&b)[8]) {
#pragma unroll
    for (int e = 0; e < 8; e++) b[e] = 0.f;
    if (bias && col < Ns) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + col);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + col + 4);
        b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
}

// Output store of the 16-bit-activation epilogues.  bf16x3 build, img > 0: C is not an fp32 activation but the [hi | hi | lo] operand IMAGE
// of the consumer GEMM (rows of 3 * img 16-bit elements, kernels.hip::k_x3_split_rows' layout and arithmetic) — the producer writes it
// directly, 6 B per element, instead of 4 B here + a 4 B read and 6 B write in the split pass.  img == 0 (always, in the 16-bit builds): the
// plain store.
__device__ __forceinline__ void epi_store8(act_t* C, int ldc, int row, int col, const float (&v)[8], int img) {
#if CC_OP == 2
    if (img > 0) {
        const uint4 hi = pack8(v);
        float h[8], d[8];
        unpack8(hi, h);
#pragma unroll
        for (int e = 0; e < 8; e++) d[e] = v[e] - h[e];
        const uint4 lo = pack8(d);
        op16_t* r3 = reinterpret_cast<op16_t*>(C) + (size_t)row * 3 * img + col;
        *reinterpret_cast<uint4*>(r3) = hi;
        *reinterpret_cast<uint4*>(r3 + img) = hi;
        *reinterpret_cast<uint4*>(r3 + 2 * img) = lo;
        return;
    }
#endif
    act_st8(C + (size_t)row * ldc + col, v);
}
// ACT / PRE >= 0 fix the run-time switches `act` / `pre` at compile time (PRE: 0 = no copy, 1 = copy, 2 = non-temporal copy): with them tested per
// 8-column unit the unrolled epilogue of the 256-row kernels is a branch tree (round 6: c_fc forward 78.2 -> 74.9 us on the 320 x 256 kernel)
template <int ACT = -1, int PRE = -1>
struct EpiBF16T {
    act_t* C;
    act_t* pre;         // nullable
    const float* bias;  // nullable
    int ldc, M, Ns;     // Ns: columns to store (multiple of 8)
    int act;            // 0 none, 1 relu, 2 gelu_new, 3 gelu_new with `pre` receiving gelu_new'(u) instead of u (the backward's multiplier)
    bool pre_nt = false; // act 3: `pre` is read only by the backward pass -> non-temporal stores
    int img = 0;         // bf16x3: > 0 = write C as the consumer's operand image with this row width (epi_store8)
    __device__ __forceinline__ int act_() const { return ACT >= 0 ? ACT : act; }
    __device__ __forceinline__ bool has_pre() const { return PRE >= 0 ? PRE != 0 : pre != nullptr; }
    __device__ __forceinline__ bool nt_() const { return PRE >= 0 ? PRE == 2 : pre_nt; }
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M || col >= Ns) return;
        if (bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + col);
            const float4 b1 = *reinterpret_cast<const float4*>(bias + col + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (act_() == 3) {
            float dg[8];
#pragma unroll
            for (int e = 0; e < 8; e++) gelu_new_both(v[e], v[e], dg[e]);
            if (has_pre()) { if (nt_()) act_st8_nt(pre + (size_t)row * ldc + col, dg); else act_st8(pre + (size_t)row * ldc + col, dg); }
        } else {
            if (has_pre()) act_st8(pre + (size_t)row * ldc + col, v);
            if (act_() == 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
            } else if (act_() == 2) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = gelu_new_f(v[e]);
            }
        }
        epi_store8(C, ldc, row, col, v, img);
    }
    // Split form for the 256-row kernels (see gemm_nt_stag256_kernel): every global LOAD of the epilogue happens before its first
    // STORE — with loads and stores both pending on the one vmcnt counter the compiler must wait vmcnt(0), i.e. drain the store
    // queue, before using any loaded value.  pre4(): per 16x16 tile in accumulator layout (4 consecutive columns of one row);
    // bias8(): the lane's 8 bias values for a column group; fin(): math + stores only.
    static constexpr bool kPre = false;
    __device__ __forceinline__ void pre4(int, int, f32x4&) const {}
    __device__ __forceinline__ void bias8(int col, float (&b)[8]) const { load_bias8(bias, col, Ns, b); }
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&b)[8]) const {
        if (row >= M || col >= Ns) return;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += b[e];
        if (act_() == 3) {
            float dg[8];
#pragma unroll
            for (int e = 0; e < 8; e++) gelu_new_both(v[e], v[e], dg[e]);
            if (has_pre()) { if (nt_()) act_st8_nt(pre + (size_t)row * ldc + col, dg); else act_st8(pre + (size_t)row * ldc + col, dg); }
        } else {
            if (has_pre()) act_st8(pre + (size_t)row * ldc + col, v);
            if (act_() == 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
            } else if (act_() == 2) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = gelu_new_f(v[e]);
            }
        }
        epi_store8(C, ldc, row, col, v, img);
    }
};
using EpiBF16 = EpiBF16T<>;

// C(bf16) = acc + bias and nothing else: EpiBF16 without its run-time switches (activation, pre-activation copy, operand image).  The 36
// input-gradient launches and the 12 c_attn launches of a GPT-2 step are this; with the switches tested per 8-column unit the epilogue of the
// 256-row kernels is an unrolled branch tree (round 6, tools/probes/q4_bench.hip: 12800 x 768 x 3072 on the 4-wave kernel 52.8 -> 49.7 us).
struct EpiBF16Plain {
    act_t* C;
    const float* bias;  // nullable
    int ldc, M, Ns;
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M || col >= Ns) return;
        if (bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + col);
            const float4 b1 = *reinterpret_cast<const float4*>(bias + col + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        act_st8(C + (size_t)row * ldc + col, v);
    }
    static constexpr bool kPre = false;
    __device__ __forceinline__ void pre4(int, int, f32x4&) const {}
    __device__ __forceinline__ void bias8(int col, float (&b)[8]) const { load_bias8(bias, col, Ns, b); }
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&b)[8]) const {
        if (row >= M || col >= Ns) return;
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += b[e];
        act_st8(C + (size_t)row * ldc + col, v);
    }
};

// out(fp32) = res + drop(acc + bias)   (residual stream update; out may alias res; drop = residual dropout, off unless drop.thresh)
struct EpiResid {
    float* out;
    const float* res;
    const float* bias;  // nullable
    int ld, M, Ns;
    Drop drop;
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M || col >= Ns) return;
        const size_t o = (size_t)row * ld + col;
        const float4 r0 = *reinterpret_cast<const float4*>(res + o);
        const float4 r1 = *reinterpret_cast<const float4*>(res + o + 4);
        float4 b0 = make_float4(0, 0, 0, 0), b1 = b0;
        if (bias) {
            b0 = *reinterpret_cast<const float4*>(bias + col);
            b1 = *reinterpret_cast<const float4*>(bias + col + 4);
        }
        float y[8] = {v[0] + b0.x, v[1] + b0.y, v[2] + b0.z, v[3] + b0.w, v[4] + b1.x, v[5] + b1.y, v[6] + b1.z, v[7] + b1.w};
        if (drop.thresh) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                float m0, m1;
                drop_mul_pair(drop, (unsigned)o + e, m0, m1);
                y[e] *= m0; y[e + 1] *= m1;
            }
        }
        *reinterpret_cast<float4*>(out + o) = make_float4(r0.x + y[0], r0.y + y[1], r0.z + y[2], r0.w + y[3]);
        *reinterpret_cast<float4*>(out + o + 4) = make_float4(r1.x + y[4], r1.y + y[5], r1.z + y[6], r1.w + y[7]);
    }
    // 256-row kernels, no dropout: the accumulators START from the residual (gemm_stag256_body), pre4 then has nothing to add
    bool acc_init = false;
    __device__ __forceinline__ bool init_from_input() const { return acc_init; }
    __device__ __forceinline__ void init4(int row, int col, f32x4& a) const {
        if (row >= M || col >= Ns) return;
        const float4 r = *reinterpret_cast<const float4*>(res + (size_t)row * ld + col);
        a[0] = r.x; a[1] = r.y; a[2] = r.z; a[3] = r.w;
    }
    static constexpr bool kPre = true;
    __device__ __forceinline__ void pre4(int row, int col, f32x4& a) const {        // acc += residual (with dropout: bias and mask too)
        if (acc_init || row >= M || col >= Ns) return;
        const size_t o = (size_t)row * ld + col;
        const float4 r = *reinterpret_cast<const float4*>(res + o);
        if (drop.thresh) {
            float4 b = make_float4(0, 0, 0, 0);
            if (bias) b = *reinterpret_cast<const float4*>(bias + col);
            float m0, m1, m2, m3;
            drop_mul_pair(drop, (unsigned)o, m0, m1);
            drop_mul_pair(drop, (unsigned)o + 2, m2, m3);
            a[0] = r.x + m0 * (a[0] + b.x);
            a[1] = r.y + m1 * (a[1] + b.y);
            a[2] = r.z + m2 * (a[2] + b.z);
            a[3] = r.w + m3 * (a[3] + b.w);
            return;
        }
        a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += r.w;
    }
    __device__ __forceinline__ void bias8(int col, float (&b)[8]) const { load_bias8(drop.thresh ? nullptr : bias, col, Ns, b); }
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&b)[8]) const {
        if (row >= M || col >= Ns) return;
        const size_t o = (size_t)row * ld + col;
        *reinterpret_cast<float4*>(out + o) = make_float4(v[0] + b[0], v[1] + b[1], v[2] + b[2], v[3] + b[3]);
        *reinterpret_cast<float4*>(out + o + 4) = make_float4(v[4] + b[4], v[5] + b[5], v[6] + b[6], v[7] + b[7]);
    }
};

// C(fp32) = / += / atomic+= alpha*acc (+bias)   (logits for the parity API, weight gradients with split-K)
struct EpiF32 {
    float* C;
    const float* bias;  // nullable (mode 0 only)
    int ldc, M, Ns;
    int mode;  // 0 store, 1 add, 2 atomic add, 3 store into the split-K slab of this blockIdx.z
    float alpha;
    size_t zstride = 0;
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M || col >= Ns) return;
        float* p = C + (size_t)row * ldc + col + (mode == 3 ? blockIdx.z * zstride : 0);
        if (mode == 2) {
#pragma unroll
            for (int e = 0; e < 8; e++) __hip_atomic_fetch_add(p + e, alpha * v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        float4 o0 = make_float4(alpha * v[0], alpha * v[1], alpha * v[2], alpha * v[3]);
        float4 o1 = make_float4(alpha * v[4], alpha * v[5], alpha * v[6], alpha * v[7]);
        if (mode == 0 && bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(bias + col);
            const float4 b1 = *reinterpret_cast<const float4*>(bias + col + 4);
            o0.x += b0.x; o0.y += b0.y; o0.z += b0.z; o0.w += b0.w; o1.x += b1.x; o1.y += b1.y; o1.z += b1.z; o1.w += b1.w;
        }
        if (mode == 1) {
            const float4 c0 = *reinterpret_cast<const float4*>(p);
            const float4 c1 = *reinterpret_cast<const float4*>(p + 4);
            o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w; o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
        }
        *reinterpret_cast<float4*>(p) = o0;
        *reinterpret_cast<float4*>(p + 4) = o1;
    }
    static constexpr bool kPre = true;
    __device__ __forceinline__ void pre4(int row, int col, f32x4& a) const {        // acc = alpha * acc (+ C for mode 1)
        a[0] *= alpha; a[1] *= alpha; a[2] *= alpha; a[3] *= alpha;
        if (mode != 1 || row >= M || col >= Ns) return;
        const float4 c = *reinterpret_cast<const float4*>(C + (size_t)row * ldc + col);
        a[0] += c.x; a[1] += c.y; a[2] += c.z; a[3] += c.w;
    }
    __device__ __forceinline__ void bias8(int col, float (&b)[8]) const { load_bias8(mode == 0 ? bias : nullptr, col, Ns, b); }
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&b)[8]) const {
        if (row >= M || col >= Ns) return;
        float* p = C + (size_t)row * ldc + col + (mode == 3 ? blockIdx.z * zstride : 0);
        if (mode == 2) {
#pragma unroll
            for (int e = 0; e < 8; e++) __hip_atomic_fetch_add(p + e, v[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        *reinterpret_cast<float4*>(p) = make_float4(v[0] + b[0], v[1] + b[1], v[2] + b[2], v[3] + b[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4] + b[4], v[5] + b[5], v[6] + b[6], v[7] + b[7]);
    }
};

// C(bf16) = acc * act'(aux)   (dgrad through relu: aux = post-activation h; through gelu_new: aux = pre-activation u)
template <int ACT = -1>      // ACT >= 0 fixes `act` at compile time (the training step's launches are all act 3)
struct EpiDActT {
    act_t* C;
    const act_t* aux;
    int ldc, M, Ns;
    int act;  // 1 relu (aux = post-activation), 2 gelu_new (aux = pre-activation u), 3 multiply by aux (= gelu_new'(u) stored by the forward)
    int img = 0;   // bf16x3: > 0 = write C as the consumer's operand image (epi_store8); aux keeps the plain [M][ldc] layout
    __device__ __forceinline__ int act_() const { return ACT >= 0 ? ACT : act; }
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        if (row >= M || col >= Ns) return;
        const size_t o = (size_t)row * ldc + col;
        float a[8];
        act_ld8(aux + o, a);
        if (act_() == 1) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = a[e] > 0.f ? v[e] : 0.f;
        } else if (act_() == 3) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] *= a[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] *= gelu_new_grad(a[e]);
        }
        epi_store8(C, ldc, row, col, v, img);
    }
    typedef act_raw8 StripAux;                                                        // 128 x 128 strip epilogue: aux fetched up front
    __device__ __forceinline__ act_raw8 load_aux(int row, int col) const {
        return (row < M && col < Ns) ? act_ldraw8(aux + (size_t)row * ldc + col) : act_zero8();
    }
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8], const act_raw8& ax) const {
        if (row >= M || col >= Ns) return;
        float a[8];
        act_unpack8(ax, a);
        if (act_() == 1) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = a[e] > 0.f ? v[e] : 0.f;
        } else if (act_() == 3) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] *= a[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] *= gelu_new_grad(a[e]);
        }
        epi_store8(C, ldc, row, col, v, img);
    }
    static constexpr bool kPre = true;
    __device__ __forceinline__ void pre4(int row, int col, f32x4& a) const {        // acc *= act'(aux)
        if (row >= M || col >= Ns) return;
        float x[4];
        act_unpack4(act_ldraw4(aux + (size_t)row * ldc + col), x[0], x[1], x[2], x[3]);
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] = act_() == 1 ? (x[e] > 0.f ? a[e] : 0.f) : (act_() == 3 ? a[e] * x[e] : a[e] * gelu_new_grad(x[e]));
    }
    __device__ __forceinline__ void bias8(int, float (&b)[8]) const {
#pragma unroll
        for (int e = 0; e < 8; e++) b[e] = 0.f;
    }
    __device__ __forceinline__ void fin(int row, int col, float (&v)[8], const float (&)[8]) const {
        if (row >= M || col >= Ns) return;
        epi_store8(C, ldc, row, col, v, img);
    }
};
using EpiDAct = EpiDActT<>;
template <class E> struct epi_is_dact { static constexpr bool value = false; };
template <int A> struct epi_is_dact<EpiDActT<A>> { static constexpr bool value = true; };

// lm_head: bf16 logits + per-(row, 64-column block) softmax partials from the fp32 accumulators + exact target logit.
struct EpiLMHead {
    act_t* C;
    float* pmax;
    float* psum;           // [M][npart]
    const int* target;     // [M] token id per row (>=0)
    float* tgt_logit;      // [M]
    int ldc, M, V, npart;  // V = true vocab (columns >= V are padding: stored as 0, excluded from the partials)
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        const bool ok = row < M && col < ldc;
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (col + e >= V) v[e] = 0.f; else m = fmaxf(m, v[e]);
        }
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (col + e < V) s += __expf(v[e] - m);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (!ok) return;
        if ((col & 63) == 0) {
            const int blk = col >> 6;
            pmax[(size_t)row * npart + blk] = m;
            psum[(size_t)row * npart + blk] = (m == -INFINITY) ? 0.f : s;
        }
        const int t = target[row] - col;
        if (t >= 0 && t < 8) tgt_logit[row] = v[t];
        act_st8(C + (size_t)row * ldc + col, v);
    }
    // 256-row kernels: the lane holds columns ca..ca+7 and cb..cb+7 of `row`; lanes l, l^16, l^32, l^48 together hold the row's
    // aligned 64-column block (all lanes of the wave must call this).
    typedef int RowAux;                                                               // per-row value fetched before the first store: the target id
    __device__ __forceinline__ int load_row(int row) const { return row < M ? target[row] : -1; }
    __device__ __forceinline__ void strip(int row, int ca, int cb, float (&va)[8], float (&vb)[8], int t) const {
        constexpr float L2E = 1.4426950408889634f;
        float m, s = 0.f;
        if (__builtin_amdgcn_readfirstlane(ca | 63) < V) {          // the whole 64-column block is inside the vocabulary (wave-uniform)
            m = fmaxf(fmaxf(fmaxf(va[0], va[1]), fmaxf(va[2], va[3])), fmaxf(fmaxf(va[4], va[5]), fmaxf(va[6], va[7])));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(vb[0], vb[1]), fmaxf(vb[2], vb[3])), fmaxf(fmaxf(vb[4], vb[5]), fmaxf(vb[6], vb[7]))));
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float ml = -m * L2E;
#pragma unroll
            for (int e = 0; e < 8; e++) s += __builtin_amdgcn_exp2f(fmaf(va[e], L2E, ml)) + __builtin_amdgcn_exp2f(fmaf(vb[e], L2E, ml));
        } else {
            m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (ca + e >= V) va[e] = 0.f; else m = fmaxf(m, va[e]);
                if (cb + e >= V) vb[e] = 0.f; else m = fmaxf(m, vb[e]);
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (ca + e < V) s += __expf(va[e] - m);
                if (cb + e < V) s += __expf(vb[e] - m);
            }
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (row >= M) return;
        if ((ca & 63) == 0 && ca < ldc) {
            const int blk = ca >> 6;
            pmax[(size_t)row * npart + blk] = m;
            psum[(size_t)row * npart + blk] = (m == -INFINITY) ? 0.f : s;
        }
        if (ca < ldc) {
            if (t >= ca && t < ca + 8) tgt_logit[row] = va[t - ca];
            act_st8(C + (size_t)row * ldc + ca, va);
        }
        if (cb < ldc) {
            if (t >= cb && t < cb + 8) tgt_logit[row] = vb[t - cb];
            act_st8(C + (size_t)row * ldc + cb, vb);
        }
    }
};
// Exponential form of the lm_head outputs (bf16 build of the training path): what is stored is NOT the logit but
// E = exp(logit - cref[row]), cref[row] = the row's own target logit computed ahead of the GEMM (k_lm_tgt_ref: the same bf16 products
// in fp32, so it doubles as the loss's exact target logit), and the partials are pmax = cref[row], psum = sum of E over the block —
// k_ce_rows then returns lse = cref + log(sum E) unchanged.  The softmax gradient (softmax - onehot) w is r[row] E - w onehot with
// the per-row scalar r = exp(cref - lse) w: the lm_head input-gradient GEMM reads E as its A operand directly and applies r and the
// one-hot row in its finishing pass (gemm_nt_deepk LmFix) — the 2 GB read-modify-write pass of k_ce_dlogits over the [B*cap, Vp]
// matrix is gone, and the stored matrix is rounded to 16 bits once (E) instead of twice (logit, then gradient).  bf16 has fp32's
// exponent range, so E is representable whenever logit - target logit < 88; the exponent is clamped at 80 (a row whose target token
// has probability < e^-80 gets a distorted but finite gradient).  fp16 lacks that range: the fp16 build keeps the logit form.
struct EpiLMHeadExp {
    act_t* C;
    float* pmax;
    float* psum;           // [M][npart]
    const float* cref;     // [M]
    int ldc, M, V, npart;
    int img = 0;           // bf16x3: > 0 = C receives the [hi | hi | lo] operand image of the input-gradient GEMM (rows of 3 * img elements, epi_store8)
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        constexpr float L2E = 1.4426950408889634f;
        const bool ok = row < M && col < ldc;
        const float ml = -(row < M ? cref[row] : 0.f) * L2E;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            v[e] = (col + e < V) ? __builtin_amdgcn_exp2f(fminf(fmaf(v[e], L2E, ml), 80.f * L2E)) : 0.f;
            s += v[e];
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (!ok) return;
        if ((col & 63) == 0) {
            const int blk = col >> 6;
            pmax[(size_t)row * npart + blk] = col < V ? cref[row] : -INFINITY;
            psum[(size_t)row * npart + blk] = s;
        }
        epi_store8(C, ldc, row, col, v, img);
    }
    typedef float RowAux;                                                             // the row's reference shift
    __device__ __forceinline__ float load_row(int row) const { return row < M ? cref[row] : 0.f; }
    __device__ __forceinline__ void strip(int row, int ca, int cb, float (&va)[8], float (&vb)[8], float c) const {
        constexpr float L2E = 1.4426950408889634f;
        const float ml = -c * L2E;
        const bool inside = __builtin_amdgcn_readfirstlane(ca | 63) < V;              // whole 64-column block inside the vocabulary (wave-uniform)
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float ea = __builtin_amdgcn_exp2f(fminf(fmaf(va[e], L2E, ml), 80.f * L2E));
            const float eb = __builtin_amdgcn_exp2f(fminf(fmaf(vb[e], L2E, ml), 80.f * L2E));
            va[e] = (inside || ca + e < V) ? ea : 0.f;
            vb[e] = (inside || cb + e < V) ? eb : 0.f;
            s += va[e] + vb[e];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (row >= M) return;
        if ((ca & 63) == 0 && ca < ldc) {
            const int blk = ca >> 6;
            pmax[(size_t)row * npart + blk] = ca < V ? c : -INFINITY;
            psum[(size_t)row * npart + blk] = s;
        }
        if (ca < ldc) epi_store8(C, ldc, row, ca, va, img);
        if (cb < ldc) epi_store8(C, ldc, row, cb, vb, img);
    }
};
// Decode lm_head: fp32 logits (what the beam / sampling kernels read) + the same per-(row, 64-column block) partials (max, sum of
// exp(x - max)) from the accumulators, so the beam update needs no pass over the 64 MB logits matrix for its row statistics and can
// bound every 64-column block by its maximum (decode.hip k_beam_fused).  Columns >= V are stored as computed (the caller's matrix may
// be wider) but excluded from the partials.
struct EpiLogits {
    float* C;
    float* pmax;
    float* psum;           // [M][npart]
    int ldc, M, Ns, V, npart;
    __device__ __forceinline__ void operator()(int row, int col, float (&v)[8]) const {
        const bool ok = row < M && col < Ns;
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (col + e < V) m = fmaxf(m, v[e]);
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (col + e < V) s += __expf(v[e] - m);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (!ok) return;
        if ((col & 63) == 0) {
            const int blk = col >> 6;
            pmax[(size_t)row * npart + blk] = m;
            psum[(size_t)row * npart + blk] = (m == -INFINITY) ? 0.f : s;
        }
        float* p = C + (size_t)row * ldc + col;
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    typedef int RowAux;
    __device__ __forceinline__ int load_row(int) const { return 0; }
    __device__ __forceinline__ void strip(int row, int ca, int cb, float (&va)[8], float (&vb)[8], int) const {
        float m = -INFINITY, s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (ca + e < V) m = fmaxf(m, va[e]);
            if (cb + e < V) m = fmaxf(m, vb[e]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
        for (int e = 0; e < 8; e++) {
            if (ca + e < V) s += __expf(va[e] - m);
            if (cb + e < V) s += __expf(vb[e] - m);
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (row >= M) return;
        if ((ca & 63) == 0 && ca < Ns) {
            const int blk = ca >> 6;
            pmax[(size_t)row * npart + blk] = m;
            psum[(size_t)row * npart + blk] = (m == -INFINITY) ? 0.f : s;
        }
        if (ca < Ns) {
            float* p = C + (size_t)row * ldc + ca;
            *reinterpret_cast<float4*>(p) = make_float4(va[0], va[1], va[2], va[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(va[4], va[5], va[6], va[7]);
        }
        if (cb < Ns) {
            float* p = C + (size_t)row * ldc + cb;
            *reinterpret_cast<float4*>(p) = make_float4(vb[0], vb[1], vb[2], vb[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(vb[4], vb[5], vb[6], vb[7]);
        }
    }
};
template <> struct epi_row_strip<EpiLogits> { static constexpr bool value = true; };
template <> struct epi_row_strip<EpiLMHead> { static constexpr bool value = true; };
template <> struct epi_row_strip<EpiLMHeadExp> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------------------
// Host launcher
// ------------------------------------------------------------------------------------------------
template <class Epi>
inline int launch_gemm_s64(const op16_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, int ksplit, int nj, const Epi& epi, int* ks_eff,
                           hipStream_t st, const op16_t* Bimg = nullptr);
// bf16x3 build: the fused two-stage forms of the NT kernels (gemm_stag256_body<X3F>, gemm_nt_glds_x3f_kernel, gemm_nt_glds4x2_x3f_kernel); CC_X3_FUSED=0: A/B switch
inline bool x3_fused_on() {
    static const bool on = kX3 && !(cc_lab_env("CC_X3_FUSED") && atoi(cc_lab_env("CC_X3_FUSED")) == 0);
    return on;
}
template <class Epi>
inline int launch_gemm(int al, int bl, const op16_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K,
                       int ksplit, const Epi& epi, hipStream_t st, int tile = -2, int group_m = 0) {   // tile: -2 = process-wide mode / chooser, else as cc_gemm_tile_mode; group_m: row tiles per ordering group (0 = default)
    if (M <= 0 || N <= 0 || K <= 0) return CC_OK;
    if ((lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return CC_ERR_SHAPE;
    if (al == 0 && (K & 7)) return CC_ERR_SHAPE;
    if (bl == 0 && (K & 7)) return CC_ERR_SHAPE;
    if (al == 1 && (M & 7)) return CC_ERR_SHAPE;
    if (bl == 1 && (N & 7)) return CC_ERR_SHAPE;
    if constexpr (!epi_row_strip<Epi>::value && !epi_strip_aux<Epi>::value) {
        if (g_gemm_s64 > 0 && al == 0 && bl == 0 && (K % G_BK) == 0 && M <= 1024)     // tools/small_gemm_bench.py (CC_GEMM_S64 = 1 / 2)
            return launch_gemm_s64(A, lda, B, ldb, M, N, K, ksplit, g_gemm_s64, epi, nullptr, st);
    }
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    static const int env_group = []() { const char* e = cc_lab_env("CC_GROUP_M"); return e ? atoi(e) : 0; }();
    g.group_m = env_group > 0 ? env_group : (group_m > 0 ? group_m : 8);
    static const int env_stagger = []() { const char* e = cc_lab_env("CC_GEMM_STAGGER"); return e ? atoi(e) : 0; }();   // tuning knob
    g.stagger = env_stagger;
    if (ksplit < 1) ksplit = 1;
    int kt = (K + G_BK - 1) / G_BK;
    int per = (kt + ksplit - 1) / ksplit;
    ksplit = (kt + per - 1) / per;
    g.k_chunk = per * G_BK;
    dim3 grid(((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN), 1, ksplit);
    // Tile choice for NT launches.  The 256-row kernel moves 1/2 (256 wide) or 2/3 (192 wide) of the 128 x 128 kernel's L2->LDS
    // bytes per flop and runs ~1.2x / ~1.15x its rate on full waves, but holds one block per CU, so wave quantisation decides:
    // cost = rounds x tile area / rate, the 128 x 128 kernel counted with 2 co-resident blocks per CU (measured table: DESIGN.md 4.1).
    // The 320 x 256 form (10 row tiles per wave, 36-KiB stages) does 1.25x the MFMAs per K-step at 1.13-1.27x the step time: it wins
    // where it saves a round (12800 x 3072: 480 tiles = 2 rounds instead of 3; the lm_head: 25 instead of 31).
    // g_gemm_tile_mode (cc_gemm_tile_mode / CC_GEMM_S256) = 0 (never) / 3 / 4 / 5 / 6 (force 256x192 / 256x256 / 320x256 / 160x256) overrides it for
    // tests and tools/gemm_tiles.py.
    const int s256 = tile != -2 ? tile : g_gemm_tile_mode;
    int nj = 0, ni = 8;
    constexpr bool can160 = !kX3 && !epi_row_strip<Epi>::value && !epi_is_dact<Epi>::value;      // the 160 x 256 form (below)
    // K slices (blockIdx.z) on the 256-row kernels only for the slab-writing fp32 epilogue and only when the caller names the tile
    constexpr bool zsplit_ok = std::is_same<Epi, EpiF32>::value;
    if (al == 0 && bl == 0 && (K % H_BK) == 0 && (ksplit == 1 || (zsplit_ok && tile > 0)) && s256 != 0) {
        const long tm = (M + H_BM - 1) / H_BM, t128 = (long)grid.x;
        const long t256 = tm * ((N + 255) / 256), t192 = tm * ((N + 191) / 192), t320 = (long)((M + 319) / 320) * ((N + 255) / 256);
        const double c128 = t128 <= 256 ? 16384.0 / 0.75 : (double)((t128 + 511) / 512) * 32768.0;
        // (bf16x3 build, round 5: crediting the 256 x 192 form with its fused two-stage loop — c192 / 1.45, which moves c_attn / c_fc forward from
        // the three-pass 256-wide forms to 3-4 rounds of fused 192-wide tiles — measured 28.55 -> 30.1 ms per step, two alternations: reverted)
        const double c256 = (double)((t256 + 255) / 256) * 65536.0 / 1.2, c192 = (double)((t192 + 255) / 256) * 49152.0 / 1.15;
        const double c320 = (double)((t320 + 255) / 256) * 81920.0 / 1.3;
        constexpr bool can192 = !epi_row_strip<Epi>::value;   // the lm_head partials assume 64-column wave strips
        // the activation-gradient epilogue (aux tile read + gelu' + store) is not hidden at one block per CU: 12800 x 3072 x 768 measured
        // 106.6 us on 320 x 256 vs 97.3 on 128 x 128 (two co-resident blocks), while the plain / gelu-forward epilogues gain (90 -> 78 us)
        // (also with the forward-stored derivative, act 3 — a single multiply —, the 128 x 128 kernel stays ahead: 12.31 vs 12.44 ms per step)
        constexpr bool can320 = !epi_is_dact<Epi>::value;
        // (round 5) 160 x 256 (5 row tiles per wave, 26-KiB stages): for the N = 768 launches at M = 12800 it is 240 tiles on the 256 CUs where
        // 256 x 192 is 200 — hipBLASLt's pick for these shapes too (MT160x256, profiles/r05_j_*).  A single-round launch takes one tile's
        // latency, so the smaller tile is the shorter launch whenever both fit one round; never for the bf16x3 build (its fused form is 192-wide).
        const long t160 = (long)((M + 159) / 160) * ((N + 255) / 256);
        const double c160 = (double)((t160 + 255) / 256) * 40960.0 / 1.1;
        static const bool s160_on = []() { const char* e = cc_lab_env("CC_GEMM_S160"); return !e || atoi(e) != 0; }();
        if (can160 && (s256 == 6 || s256 == 7 || (s256 < 0 && s160_on && ksplit == 1 && c160 < 0.98 * c128 && c160 < 0.98 * c256 && c160 < 0.98 * c320 && (!can192 || c160 < 0.98 * c192)))) { nj = 4; ni = 5; }
        else if (s256 == 5 || (can320 && s256 < 0 && c320 < 0.98 * c128 && c320 < c256 && (!can192 || c320 < c192))) { nj = 4; ni = 10; }
        else if (s256 == 4 || (s256 < 0 && c256 < 0.98 * c128 && (!can192 || c256 <= c192))) nj = 4;
        else if (can192 && (s256 == 3 || (s256 < 0 && c192 < 0.98 * c128))) nj = 3;
    }
    if (nj) {
        const int bm = 32 * ni;
        const size_t sh = (size_t)H_NS * (bm + H_BN) * H_BK * 2;
        const dim3 gr((unsigned)(((M + bm - 1) / bm) * ((N + 64 * nj - 1) / (64 * nj))), 1, (unsigned)ksplit);
#define CC_LAUNCH_STAG(NJ_, NI_)                                                                                         \
    {                                                                                                                    \
        static bool attr_ = false;                                                                                       \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<Epi, NJ_, false, NI_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr_ = true; } \
        hipLaunchKernelGGL((gemm_nt_stag256_kernel<Epi, NJ_, false, NI_>), gr, dim3(512), sh, st, A, B, g, epi);        \
    }
        if (nj == 4 && ni == 8) CC_LAUNCH_STAG(4, 8)
        else if (nj == 4 && ni == 5) {
            if constexpr (can160) {
                // (round 6) the 160 x 256 launches go to the persistent 4-wave kernel with 64-deep full-line stages (gemm_q4.hip.h) wherever K fits its
                // ring: three stages for K % 192 == 0, two for K % 128 == 0; byte offsets inside an operand are 32-bit there.  Measured against the
                // staggered 160 x 256 kernel (profiles/r06_l): 12800 x 768 x 3072 58.9 -> 52.8 us, x 2304 44.2 -> 41.7, x 768 20.2 -> 21.2 (plain functor:
                // 57.1 -> 49.7, 43.7 -> 38.5, 20.6 -> 19.8).  cc_gemm_tile_mode 6 keeps the staggered kernel, 7 forces this one.
                static const bool q4_on = []() { const char* e = cc_lab_env("CC_GEMM_Q4"); return !e || atoi(e) != 0; }();
                const bool fits32 = (size_t)M * (size_t)lda * 2 < 0xffff0000ull && (size_t)N * (size_t)ldb * 2 < 0xffff0000ull;
                const int ns = (K % 192 == 0 && K >= 384) ? 3 : ((K % 128 == 0 && K >= 256) ? 2 : 0);
                static const bool q4_last = []() { const char* e = cc_lab_env("CC_Q4_LAST"); return !e || atoi(e) != 0; }();
                g.flags = q4_last ? 0 : 1;
                if (ns && fits32 && ksplit == 1 && s256 != 6 && (q4_on || s256 == 7)) {
                    static const int ncu = []() { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
                    const int tiles = ((M + 159) / 160) * ((N + 255) / 256);
                    const dim3 gq((unsigned)(tiles < ncu ? tiles : ncu));
                    if (ns == 3) {
                        constexpr size_t shq = (size_t)3 * (160 + 256) * 128;
                        static bool attr_ = false;
                        if (!attr_) { (void)hipFuncSetAttribute((const void*)gemm_nt_q4_kernel<Epi, 5, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shq); attr_ = true; }
                        hipLaunchKernelGGL((gemm_nt_q4_kernel<Epi, 5, 3>), gq, dim3(256), shq, st, A, B, g, epi);
                    } else {
                        constexpr size_t shq = (size_t)2 * (160 + 256) * 128;
                        static bool attr_ = false;
                        if (!attr_) { (void)hipFuncSetAttribute((const void*)gemm_nt_q4_kernel<Epi, 5, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shq); attr_ = true; }
                        hipLaunchKernelGGL((gemm_nt_q4_kernel<Epi, 5, 2>), gq, dim3(256), shq, st, A, B, g, epi);
                    }
                } else CC_LAUNCH_STAG(4, 5)
            }
        }
        else if (nj == 4) CC_LAUNCH_STAG(4, 10)
        else if constexpr (!epi_row_strip<Epi>::value) {
            bool fused = false;
            if constexpr (kX3) {       // bf16x3 build: every NT launch carries operand images over K' = 3 K — the fused two-stage form (gemm_stag256_body, X3F)
                const bool x3f_on = x3_fused_on();
                if (x3f_on && (K % (3 * H_BK)) == 0 && ksplit <= K / (3 * H_BK)) {     // (K slices: ranges of the logical K's chunks, every slice non-empty)
                    static bool attr_ = false;
                    if (!attr_) { (void)hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<Epi, 3, false, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr_ = true; }
                    hipLaunchKernelGGL((gemm_nt_stag256_kernel<Epi, 3, false, 8, true>), gr, dim3(512), sh, st, A, B, g, epi);
                    fused = true;
                }
            }
            if (!fused) CC_LAUNCH_STAG(3, 8)
        }
#undef CC_LAUNCH_STAG
    } else
#ifdef CC_GEMM_ABLATION
    static const int abl = []() { const char* e = cc_lab_env("CC_GEMM_ABL"); return e ? atoi(e) : 0; }();
    if (al == 0 && bl == 0 && (K % G_BK) == 0 && abl == 1) { hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, 1>), grid, dim3(G_THREADS), 0, st, A, B, g, epi); }
    else if (al == 0 && bl == 0 && (K % G_BK) == 0 && abl == 2) { hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, 2>), grid, dim3(G_THREADS), 0, st, A, B, g, epi); }
    else if (al == 0 && bl == 0 && (K % G_BK) == 0 && abl == 4) { hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, 4>), grid, dim3(G_THREADS), 0, st, A, B, g, epi); }
    else if (al == 0 && bl == 0 && (K % G_BK) == 0 && abl == 5) { hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, 5>), grid, dim3(G_THREADS), 0, st, A, B, g, epi); }
    else if (al == 0 && bl == 0 && (K % G_BK) == 0 && abl == 6) { hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, 6>), grid, dim3(G_THREADS), 0, st, A, B, g, epi); }
    else
#endif
    if (al == 0 && bl == 0 && (K % G_BK) == 0 && (long)grid.x * grid.z <= 256 && g_gemm_tile_mode != 0 && K / (int)grid.z >= 4 * G_BK) {
        // at most one block per CU: nothing co-resident to hide the 2-stage kernel's per-K-step round trip -> 4-stage variant
        constexpr size_t sh4 = (size_t)8 * G_TILE_BYTES;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)gemm_nt_glds4_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
            (void)hipFuncSetAttribute((const void*)gemm_nt_glds4x2_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4);
            attr = true;
        }
        bool fused = false;
        if constexpr (kX3) {
            const bool x3f_on = x3_fused_on();
            if (g_gemm_small_x2 && x3f_on && ksplit == 1 && (K % (3 * G_BK)) == 0) {
                static bool attr2 = false;
                if (!attr2) { (void)hipFuncSetAttribute((const void*)gemm_nt_glds4x2_x3f_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh4); attr2 = true; }
                hipLaunchKernelGGL((gemm_nt_glds4x2_x3f_kernel<Epi>), grid, dim3(2 * G_THREADS), sh4, st, A, B, g, epi);
                fused = true;
            }
        }
        if (fused) {}
        else if (g_gemm_small_x2) hipLaunchKernelGGL((gemm_nt_glds4x2_kernel<Epi>), grid, dim3(2 * G_THREADS), sh4, st, A, B, g, epi);
        else hipLaunchKernelGGL((gemm_nt_glds4_kernel<Epi>), grid, dim3(G_THREADS), sh4, st, A, B, g, epi);
    } else if (al == 0 && bl == 0 && (K % G_BK) == 0) {
        bool fused = false;
        if constexpr (kX3) {       // bf16x3 build: operand images over K' = 3 K -> the fused two-stage form
            const bool x3f_on = x3_fused_on();
            if (x3f_on && ksplit == 1 && (K % (3 * G_BK)) == 0) {
                hipLaunchKernelGGL((gemm_nt_glds_x3f_kernel<Epi>), grid, dim3(G_THREADS), 0, st, A, B, g, epi);
                fused = true;
            }
        }
        if (!fused) hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi>), grid, dim3(G_THREADS), 0, st, A, B, g, epi);
    }
    else if (al == 0 && bl == 0)
        hipLaunchKernelGGL((gemm_bf16_kernel<0, 0, Epi>), grid, dim3(G_THREADS), 0, st, A, B, g, epi);
    else if (al == 0 && bl == 1)
        hipLaunchKernelGGL((gemm_bf16_kernel<0, 1, Epi>), grid, dim3(G_THREADS), 0, st, A, B, g, epi);
    else if (al == 1 && bl == 1)
        hipLaunchKernelGGL((gemm_bf16_kernel<1, 1, Epi>), grid, dim3(G_THREADS), 0, st, A, B, g, epi);
    else
        return CC_ERR_ARG;
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

// Skinny NT launcher (gemm_nt_s64_kernel): K % 64 == 0; nj = 1 (64 x 64 tiles), 2 (64 x 128), 3 (64 x 64, K over the waves) or 4 (80 x 64, K over the waves); K split over blockIdx.z
// (epi must be an EpiF32 in slab mode when ksplit > 1).  *ks_eff returns the effective slice count.
template <class Epi>
inline int launch_gemm_s64(const op16_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, int ksplit, int nj, const Epi& epi, int* ks_eff,
                           hipStream_t st, const op16_t* Bimg) {
    if ((K % G_BK) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return CC_ERR_SHAPE;
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.group_m = 8;
    g.stagger = 0;
    if (ksplit < 1) ksplit = 1;
    const int kt = K / G_BK;
    const int per = (kt + ksplit - 1) / ksplit;
    ksplit = (kt + per - 1) / per;
    g.k_chunk = per * G_BK;
    if (ks_eff) *ks_eff = ksplit;
    const int tm = (M + 63) / 64;
#define S64_LAUNCH(NJ_, NS_, KG_)                                                                                        \
    {                                                                                                                    \
        constexpr size_t sh = (size_t)(NS_) * (64 + 64 * (NJ_)) * 128;                                                   \
        static bool attr = false;                                                                                        \
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_s64_kernel<Epi, NJ_, NS_, KG_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
        hipLaunchKernelGGL((gemm_nt_s64_kernel<Epi, NJ_, NS_, KG_>), dim3((unsigned)(tm * ((N + 64 * (NJ_) - 1) / (64 * (NJ_)))), 1, (unsigned)ksplit), \
                           dim3(G_THREADS * (KG_)), sh, st, A, B, g, epi);                                               \
    }
#ifdef CC_EXPERIMENTS
    if (nj == 3 && Bimg && (N % 64) == 0 && ((uintptr_t)Bimg & 15) == 0) {      // ... with the weight operand global -> VGPR from its fragment-ordered image
        constexpr size_t sh = (size_t)4 * 128 * 128;      // (the epilogue's partial tiles need the 64 KiB)
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_s64kwb_kernel<Epi, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
        hipLaunchKernelGGL((gemm_nt_s64kwb_kernel<Epi, 4>), dim3((unsigned)(tm * (N / 64)), 1, (unsigned)ksplit), dim3(320), sh, st, A, Bimg, g, epi);
    } else
#else
    (void)Bimg;
#endif
    if (nj == 4) {                   // 80 x 64 tiles, K split over the waves (gemm_nt_s80kw_kernel)
        constexpr size_t sh = (size_t)4 * 6 * 16 * 64 * sizeof(float);      // the epilogue's partial tiles: 96 KiB (the four 20-KiB stages fit inside)
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_s80kw_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
        hipLaunchKernelGGL((gemm_nt_s80kw_kernel<Epi>), dim3((unsigned)(((M + 79) / 80) * ((N + 63) / 64)), 1, (unsigned)ksplit), dim3(G_THREADS), sh, st, A, B, g, epi);
    } else if (nj == 3) {                   // K split over the waves (gemm_nt_s64kw_kernel)
        constexpr size_t sh = (size_t)4 * 128 * 128;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_s64kw_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
        hipLaunchKernelGGL((gemm_nt_s64kw_kernel<Epi>), dim3((unsigned)(tm * ((N + 63) / 64)), 1, (unsigned)ksplit), dim3(G_THREADS), sh, st, A, B, g, epi);
    } else if (nj == 2) S64_LAUNCH(2, 3, 1) else S64_LAUNCH(1, 4, 1)
#undef S64_LAUNCH
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

// Weight-gradient form C[M][N] = sum_k A[k][M] B[k][N] (both operands K-strided) on the 256 x 256 kernel with direct-to-LDS staging
// and transpose reads; K is split over blockIdx.z (epi must be an EpiF32 in slab mode when ksplit > 1).  K % 32 == 0, M % 8 == 0,
// N % 8 == 0.  Returns the effective slice count through *ks_eff.
template <class Epi>
inline int launch_gemm_tt256(const op16_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, int ksplit, const Epi& epi, int* ks_eff,
                             hipStream_t st) {
    if ((K % H_BK) || (M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return CC_ERR_SHAPE;
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.group_m = 8;
    g.stagger = 0;
    if (ksplit < 1) ksplit = 1;
    const int kt = (K + G_BK - 1) / G_BK;
    const int per = (kt + ksplit - 1) / ksplit;
    ksplit = (kt + per - 1) / per;
    g.k_chunk = per * G_BK;
    if (ks_eff) *ks_eff = ksplit;
    constexpr size_t sh = (size_t)H_NS * H_STAGE;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_stag256_kernel<Epi, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const dim3 gr((unsigned)(((M + H_BM - 1) / H_BM) * ((N + H_BN - 1) / H_BN)), 1, (unsigned)ksplit);
    hipLaunchKernelGGL((gemm_nt_stag256_kernel<Epi, 4, true>), gr, dim3(512), sh, st, A, B, g, epi);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

// Grouped form: up to 32 independent weight-gradient problems (one mapper / GPT-2 layer's dW's, or — round 5 — all layers of a mapper backward) in ONE launch.  The blocks of all
// problems and all their K slices form one 1-D grid (problem i owns logical blocks [first[i], first[i+1]), slice-major inside), so the
// launch fills the CUs with ONE tail instead of four, and the per-launch floor is paid once.  Each problem writes fp32 slabs (C of
// its functor + slice * zstride) that the batched slab reduce folds into dW.
constexpr int TT_GROUP_MAX = 32;
struct TTGroup {
    const op16_t* A[TT_GROUP_MAX];
    const op16_t* B[TT_GROUP_MAX];
    GemmShape g[TT_GROUP_MAX];
    float* slab[TT_GROUP_MAX];        // slab base of problem i (slice z at + z * zstride[i]); direct mode: dW itself
    size_t zstride[TT_GROUP_MAX];
    int ldc[TT_GROUP_MAX];            // row stride of the output (slab: g[i].N; direct: ldw)
    int first[TT_GROUP_MAX + 1];      // logical block ranges
    int n;
    int mode;                         // EpiF32 mode of every problem: 0 = store into the slab, 1 = dW += tile (one K slice, no slab, no reduce)
    // direct mode, 256 x 256 kernel: the tiles beyond the last FULL round of the chip (physical blocks >= `whole`, dispatched last) are cut into
    // `split` K slices each, so the launch ends with a short round of all CUs instead of a long round of a few (8 mapper layers: 576 tiles =
    // 2 full rounds + 64 tiles x 4 slices).  The slices store compact 256 x 256 fp32 slabs into `tail` (unit u at + u * 65536) and
    // gemm_tt_tail_reduce_kernel adds them into dW (fp32 atomics for the same job measured 2.54 -> 2.84 ms on the mapper).  whole < 0: off.
    int whole = -1, split = 1;
    float* tail = nullptr;
};
__device__ __forceinline__ int tt_group_find(const TTGroup& grp, int L) {
    int i = 0;
    for (int k = 1; k < grp.n; k++)
        if (L >= grp.first[k]) i = k;
    return i;
}
static __global__ __launch_bounds__(G_THREADS, 1) void gemm_tt_glds4_group_kernel(TTGroup grp) {
    extern __shared__ __attribute__((aligned(1024))) char smemt[];
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int i = tt_group_find(grp, L);
    const GemmShape g = grp.g[i];
    const int tiles = ((g.M + G_BM - 1) / G_BM) * ((g.N + G_BN - 1) / G_BN);
    const int local = L - grp.first[i];
    const int z = local / tiles, tile = local - z * tiles;
    EpiF32 e{grp.slab[i] + (size_t)z * grp.zstride[i], nullptr, grp.ldc[i], g.M, g.N, grp.mode, 1.0f};
    gemm_tt_glds4_body(grp.A[i], grp.B[i], g, e, tile, z, smemt);
}
// the same grouping on the 256 x 256 transpose-read kernel (8 waves, one block per CU): fewer, fatter blocks — wins when the group's
// tiles x slices fill the CUs in one round (a mapper layer: 72 tiles x 3 slices)
static __global__ __launch_bounds__(512, 1) void gemm_tt_stag256_group_kernel(TTGroup grp) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    if (grp.whole >= 0) {
        // direct mode with a K-sliced tail: logical tile index from the PHYSICAL block index (dispatch order decides what runs last)
        const int b = (int)blockIdx.x;
        const bool cut = b >= grp.whole;
        const int Lt = cut ? grp.whole + (b - grp.whole) / grp.split : xcd_remap(b, grp.whole);
        const int i = tt_group_find(grp, Lt);
        GemmShape g = grp.g[i];
        int z = 0;
        if (cut) {
            z = (b - grp.whole) % grp.split;
            const int kt = (g.K + G_BK - 1) / G_BK;
            g.k_chunk = ((kt + grp.split - 1) / grp.split) * G_BK;
        }
        if (!cut) {
            EpiF32 e{grp.slab[i], nullptr, grp.ldc[i], g.M, g.N, 1, 1.0f};
            gemm_stag256_body<EpiF32, 4, true>(grp.A[i], grp.B[i], g, e, Lt - grp.first[i], 0, smem);
        } else {
            // compact slab of this (tile, slice): element (row, col) of the tile at (row - m0) * 256 + (col - n0)
            const int tiles_n = (g.N + H_BN - 1) / H_BN, tiles_m = (g.M + H_BM - 1) / H_BM;
            int tm, tn;
            tile_coords(Lt - grp.first[i], tiles_m, tiles_n, g.group_m, tm, tn);
            float* sl = grp.tail + (size_t)(b - grp.whole) * (H_BM * H_BN) - ((size_t)tm * H_BM * H_BN + (size_t)tn * H_BN);
            EpiF32 e{sl, nullptr, H_BN, g.M, g.N, 0, 1.0f};
            gemm_stag256_body<EpiF32, 4, true>(grp.A[i], grp.B[i], g, e, Lt - grp.first[i], z, smem);
        }
        return;
    }
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int i = tt_group_find(grp, L);
    const GemmShape g = grp.g[i];
    const int tiles = ((g.M + H_BM - 1) / H_BM) * ((g.N + H_BN - 1) / H_BN);
    const int local = L - grp.first[i];
    const int z = local / tiles, tile = local - z * tiles;
    EpiF32 e{grp.slab[i] + (size_t)z * grp.zstride[i], nullptr, grp.ldc[i], g.M, g.N, grp.mode, 1.0f};
    gemm_stag256_body<EpiF32, 4, true>(grp.A[i], grp.B[i], g, e, tile, z, smem);
}
// dW += sum of the K-slice slabs of the cut tiles (16 workgroups per cut tile, 16 rows each: with one per tile the 64-workgroup launch ran at 0.8 TB/s)
static __global__ __launch_bounds__(256) void gemm_tt_tail_reduce_kernel(TTGroup grp) {
    const int ct = (int)blockIdx.x >> 4, rb = ((int)blockIdx.x & 15) * (H_BM / 16);
    const int Lt = grp.whole + ct;
    const int i = tt_group_find(grp, Lt);
    const GemmShape g = grp.g[i];
    const int tiles_n = (g.N + H_BN - 1) / H_BN, tiles_m = (g.M + H_BM - 1) / H_BM;
    int tm, tn;
    tile_coords(Lt - grp.first[i], tiles_m, tiles_n, g.group_m, tm, tn);
    const int m0 = tm * H_BM, n0 = tn * H_BN;
    const float* sl = grp.tail + (size_t)ct * grp.split * (H_BM * H_BN);
    float* dW = grp.slab[i];
    const int ldw = grp.ldc[i];
    for (int q = threadIdx.x; q < (H_BM / 16) * H_BN / 4; q += 256) {
        const int r = rb + q / (H_BN / 4), c = (q % (H_BN / 4)) * 4;
        if (m0 + r >= g.M || n0 + c >= g.N) continue;
        float4 a = *reinterpret_cast<const float4*>(sl + (size_t)r * H_BN + c);
        for (int z = 1; z < grp.split; z++) {
            const float4 v = *reinterpret_cast<const float4*>(sl + (size_t)z * (H_BM * H_BN) + (size_t)r * H_BN + c);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        float* d = dW + (size_t)(m0 + r) * ldw + n0 + c;
        const float4 o = *reinterpret_cast<const float4*>(d);
        *reinterpret_cast<float4*>(d) = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
    }
}
inline int launch_gemm_tt256_group(const TTGroup& grp, hipStream_t st) {
    constexpr size_t sh = (size_t)H_NS * H_STAGE;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tt_stag256_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const unsigned blocks = grp.whole >= 0 ? (unsigned)(grp.whole + (grp.first[grp.n] - grp.whole) * grp.split) : (unsigned)grp.first[grp.n];
    hipLaunchKernelGGL(gemm_tt_stag256_group_kernel, dim3(blocks), dim3(512), sh, st, grp);
    if (grp.whole >= 0) hipLaunchKernelGGL(gemm_tt_tail_reduce_kernel, dim3((unsigned)(grp.first[grp.n] - grp.whole) * 16), dim3(256), 0, st, grp);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}
inline int launch_gemm_tt128_group(const TTGroup& grp, hipStream_t st) {
    const size_t sh = 4 * 2 * G_TILE_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tt_glds4_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    hipLaunchKernelGGL(gemm_tt_glds4_group_kernel, dim3((unsigned)grp.first[grp.n]), dim3(G_THREADS), sh, st, grp);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

// TT 128 x 128 launcher (see gemm_tt_glds4_kernel): K % 64 == 0, M % 8 == 0, N % 8 == 0; slices over blockIdx.z.
template <class Epi>
inline int launch_gemm_tt128(const op16_t* A, int lda, const op16_t* B, int ldb, int M, int N, int K, int ksplit, const Epi& epi, int* ks_eff,
                             hipStream_t st) {
    if ((K % G_BK) || (M & 7) || (N & 7) || (lda & 7) || (ldb & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return CC_ERR_SHAPE;
    GemmShape g;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.group_m = 8;
    g.stagger = 0;
    if (ksplit < 1) ksplit = 1;
    const int kt = K / G_BK;
    const int per = (kt + ksplit - 1) / ksplit;
    ksplit = (kt + per - 1) / per;
    g.k_chunk = per * G_BK;
    if (ks_eff) *ks_eff = ksplit;
    constexpr size_t sh = (size_t)8 * G_TILE_BYTES;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tt_glds4_kernel<Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
    const dim3 gr((unsigned)(((M + G_BM - 1) / G_BM) * ((N + G_BN - 1) / G_BN)), 1, (unsigned)ksplit);
    hipLaunchKernelGGL((gemm_tt_glds4_kernel<Epi>), gr, dim3(G_THREADS), sh, st, A, B, g, epi);
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_LAUNCH;
}

}  // namespace CC_NS
