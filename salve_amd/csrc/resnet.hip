// resnet.hip -- early-fusion ResNet verifier forward pass for gfx950 (MI355X), fp16 MFMA with fp32 accumulation.
//
// Stands behind salve/models/early_fusion.py:41-83 (EarlyFusionCEResnet.forward) with the torchvision ResNet v1.5
// trunk built by salve/models/resnet_factory.py:26-44.  The host (salve_amd/models) folds every BatchNorm into the
// preceding convolution, packs weights as [Cout][KH][KWp][Cin] fp16 and hands the network over as a small "op
// program" (salve_resnet_op_t): CONV (+bias, +residual, ReLU), MAXPOOL 3x3/2, AVGPOOL+FC.  The library just runs it.
//
// Convolution = implicit GEMM on NHWC fp16 activations:  out[m, n] = sum_k A[m, k] W[n, k],  m = (b, oy, ox),
// k = (kh, kw, ci).  Block tile 128 x BN x 64, four waves (2 x 2), v_mfma_f32_16x16x32_f16, the im2col gather of
// A staged global -> registers -> LDS one k-tile ahead of the MFMAs (issue-early / write-late), epilogue through
// LDS so that the residual read and the output store are 16-byte coalesced.  The 7x7/2 stem runs through the same
// kernel: input channels are padded to 8 (or 16) and kw to 8 so that one k-tile is one kernel row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>

#include "../../include/salve_hip.h"
#include "salve_common.h"

namespace {

typedef __attribute__((__ext_vector_type__(8))) _Float16 act8;  // 8 activations / weights: IEEE half precision
typedef __attribute__((__ext_vector_type__(4))) float f32x4;

constexpr int BM = 128;
constexpr int BK = 64;       // one k-tile = 64 halves = one 128-byte LDS row = 8 chunks of 16 bytes
constexpr int CONV_THREADS = 256;

struct ConvArgs {
    const uint16_t* in;
    const uint16_t* w;
    const float* bias;
    const uint16_t* res;
    uint16_t* out;
    const int32_t* ktab;    // per 8-element chunk of K: dy | dx << 8 | channel offset << 16
    const uint16_t* zeros;  // >= 16 zero bytes: the source of padding taps and of rows beyond M
    int B, Hi, Wi, Cin, Ho, Wo, Cout, stride, pad, K, M, relu;
    int m_tiles, n_tiles;
    // second, point-wise source (SRC2 kernels): k-tiles from nkt1 on read Cin2 channels of in2 at (oy, ox) * stride2
    const uint16_t* in2;
    int Hi2, Wi2, Cin2, stride2, nkt1;
    int32_t* status;        // device status word (salve_hip.h) or nullptr
    int KW, cin_log2;       // conv_wide_kernel: padded kernel width, log2(Cin) (Cin is a power of two there)
    int xcd_contig;         // workgroup -> tile mapping: 1 = every XCD owns a contiguous range of m-tiles (xcd_tile below)
};

// Workgroup -> tile, XCD-aware.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with an L2 of its own), so
// XCD x runs ids x, x + 8, x + 16, ... in that order.  `contig`: XCD x owns the m-tiles [x * mper, (x + 1) * mper), walks them
// in order and runs the n-tiles of one m-tile back to back -- the n-tiles share their activation rows through that L2, AND
// consecutive m-tiles (the neighbouring image rows a 3x3 gather reads again) stay in the same L2.  Measured (round 3, PMC
// FETCH_SIZE at batch 4096): with m-tiles dealt round-robin instead (m_tile % 8 = XCD, the round-1 mapping) the 3x3 kernels
// fetched 1.8 - 2.1 x their input from HBM and the stride-2 3x3 of layer 2 ran AT the HBM roof (6.5 TB/s) on re-reads.
// The grid is 8 * mper * n_tiles workgroups, mper = ceil(m_tiles / 8); workgroups beyond the last m-tile leave at once.
__device__ __forceinline__ bool xcd_tile(int id, int m_tiles, int n_tiles, int contig, int& m_tile, int& n_tile) {
    if (contig) {
        const int mper = (m_tiles + 7) >> 3;
        const int xcd = id & 7, s = id >> 3;
        n_tile = s % n_tiles;
        m_tile = xcd * mper + s / n_tiles;
        return s / n_tiles < mper && m_tile < m_tiles;
    }
    const int per_group = 8 * n_tiles;
    const int g = id / per_group, r = id % per_group;
    m_tile = g * 8 + (r & 7);
    n_tile = r >> 3;
    return m_tile < m_tiles;
}
// The same for a one-dimensional list of `n` tiles (fused bottleneck blocks, stem strips): XCD x owns a contiguous range, so
// the tiles of one image -- which share halo rows -- meet in one L2.  Bijective for every n (the guide's form).
__device__ __forceinline__ int xcd_linear(int id, int n, int contig) {
    if (!contig) return id;
    const int q = n >> 3, r = n & 7, xcd = id & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
}

// Activations and weights are IEEE half precision (fp16: 11 significand bits; the MFMA rate is that of bf16).  With bf16
// storage (8 bits) the logits of the 152-layer network missed the 1e-3 parity bound; fp16 meets it with margin.  The
// conversion saturates at the fp16 range instead of producing infinities.
__device__ __forceinline__ float act_to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f32_to_act(float f) {
    f = fminf(fmaxf(f, -65504.f), 65504.f);  // (v_max / v_min return the non-NaN operand: a NaN input comes out as -65504)
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}
// Saturation must not pass silently: every epilogue folds the magnitudes it stores into `amax` (v_max3_f32 with |.|
// source modifiers: half an instruction per value) and reports once per thread when the fp16 range was exceeded.
// NaN: the hardware maxima used here and in the ReLUs return their non-NaN operand, so a NaN activation is turned into 0 by
// the next ReLU instead of propagating to the logits as it does in torch -- and does NOT set the status bit.  A NaN can enter
// only through the network input or the weights (fp16 products accumulated in fp32 do not overflow): the host refuses
// non-finite weights when it packs them (hip_resnet.build_program), the rasteriser's tiles are finite by construction, and
// a caller that hands its own tensors to EarlyFusionCEResnet.forward is told so there.
__device__ __forceinline__ void track4(float& amax, float v0, float v1, float v2, float v3) {
    amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
    amax = fmaxf(amax, fmaxf(fabsf(v2), fabsf(v3)));
}
// The epilogue arithmetic of every convolution kernel: four fp32 sums (accumulator + bias [+ residual]) -> [ReLU] -> saturate ->
// fp16, two per dword.  RELU: v_max3 (range tracking; negative sums cannot raise a maximum that starts at 0 and the ReLU stores
// 0 for them) + 2 v_med3 (ReLU and saturation in one: the median of x, 0, 65504) + v_cvt_pk_f16_f32 per PAIR -- 2 instructions
// per value where fmaxf / fminf / two conversions / or took 5 (round 3: the fused 56x56 block issued 1250 vector instructions
// per wave and tile next to its 152 MFMAs).  Same values as f32_to_act(fmaxf(v, 0)): one rounding, to nearest even; a NaN sum
// stores 0 (-65504 without ReLU), as before.
template <bool RELU>
__device__ __forceinline__ uint32_t pack2(float& amax, float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    constexpr float lo = RELU ? 0.f : -65504.f;
    amax = RELU ? fmaxf(amax, fmaxf(a, b)) : fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
    const f2 v = {__builtin_amdgcn_fmed3f(a, lo, 65504.f), __builtin_amdgcn_fmed3f(b, lo, 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}
template <bool RELU>
__device__ __forceinline__ uint2 pack4(float& amax, f32x4 v) {
    uint2 o;
    o.x = pack2<RELU>(amax, v[0], v[1]);
    o.y = pack2<RELU>(amax, v[2], v[3]);
    return o;
}
// The same for the general kernels, whose ReLU is a launch argument: `lo` = 0 (ReLU) or -65504, the low side of the range tracked
// as a minimum of its own (half an instruction more per value) -- ONE code path: two copies of the epilogue under a branch made
// hipcc spill in conv8_kernel, whose 128 accumulator registers are live there.  Report fmaxf(amax, relu ? 0 : -amin).
__device__ __forceinline__ uint2 pack4_lo(float& amax, float& amin, f32x4 v, float lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    amax = fmaxf(amax, fmaxf(v[0], v[1]));
    amax = fmaxf(amax, fmaxf(v[2], v[3]));
    amin = fminf(amin, fminf(v[0], v[1]));
    amin = fminf(amin, fminf(v[2], v[3]));
    const f2 a = {__builtin_amdgcn_fmed3f(v[0], lo, 65504.f), __builtin_amdgcn_fmed3f(v[1], lo, 65504.f)};
    const f2 b = {__builtin_amdgcn_fmed3f(v[2], lo, 65504.f), __builtin_amdgcn_fmed3f(v[3], lo, 65504.f)};
    uint2 o;
    o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, h2));
    o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, h2));
    return o;
}
// The sums as vector expressions, so that they become v_pk_add_f32 (two fp32 additions per instruction; same IEEE results).
__device__ __forceinline__ f32x4 vec4(float4 b) { return f32x4{b.x, b.y, b.z, b.w}; }
__device__ __forceinline__ f32x4 vec4(uint2 r) {   // four fp16 residual values
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    return __builtin_convertvector(__builtin_bit_cast(h4, r), f32x4);
}
__device__ __forceinline__ void report_range(int32_t* status, float amax) {
    if (status && !(amax <= 65504.f)) atomicOr(status, SALVE_STATUS_FP16_RANGE);
}

typedef __attribute__((address_space(1))) const void* global_cptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// LDS accesses next to LDS-DMA in flight.  hipcc (ROCm 7.2) puts an `s_waitcnt vmcnt(0)` in front of every LDS access it takes
// for a possible alias of an LDS-DMA destination -- which drains a prefetch that is meant to stay in flight -- and decides by
// the access TYPE: loads and stores of _Float16 vectors (the MFMA fragments) are left alone, the same bytes accessed as uint2 /
// uint4 / float get the wait (checked on small kernels and in the ISA of these, round 3; the fragment reads of every kernel
// here have always been exempt, which is why the counted waits of conv8_kernel work at all).  The helpers below keep an access
// typed <n x half> whatever its bits mean: the empty asm makes the value opaque, so the optimiser cannot re-type the load or
// store from its uses.  Ordering against the LDS-DMA is the caller's business (counted vmcnt + barrier), as for the fragments.
typedef __attribute__((__ext_vector_type__(4))) _Float16 half4v;
__device__ __forceinline__ uint2 lds_read8(const uint16_t* p) {
    half4v t = *reinterpret_cast<const half4v*>(p);
    asm volatile("" : "+v"(t));
    return __builtin_bit_cast(uint2, t);
}
__device__ __forceinline__ void lds_write8(uint16_t* p, uint2 v) {
    half4v t = __builtin_bit_cast(half4v, v);
    asm volatile("" : "+v"(t));
    *reinterpret_cast<half4v*>(p) = t;
}
__device__ __forceinline__ uint4 lds_read16(const uint16_t* p) {
    act8 t = *reinterpret_cast<const act8*>(p);
    asm volatile("" : "+v"(t));
    return __builtin_bit_cast(uint4, t);
}
__device__ __forceinline__ float4 lds_read_f4(const float* p) {
    act8 t = *reinterpret_cast<const act8*>(p);
    asm volatile("" : "+v"(t));
    return __builtin_bit_cast(float4, t);
}

// Staging: `global_load_lds_dwordx4` -- each lane names its own 16 source bytes (the im2col gather), the wave's 1 KiB
// lands lane-linearly in LDS without passing through VGPRs or the ds_write path.  One wave-instruction fills 8 tile
// rows of 128 bytes, so the LDS image is unpadded; bank conflicts of the 16-byte fragment reads are avoided by a swizzle
// applied on the SOURCE side: LDS slot (row r, 16-byte slot q) holds k-chunk q ^ ((r >> 1) & 7).
// One LDS stage (load, barrier, multiply, barrier; <= 35 KB): four workgroups per CU hide each other's latency.  Measured
// on every layer of ResNet-50 at batch 512, that beats two stages with the next tile's loads in flight under the MFMAs
// (64 KB, two workgroups per CU) by 25-45 %: occupancy, not explicit pipelining, is what this tile size wants.
// POINTWISE: 1x1 / stride 1 / pad 0 -- output pixel m reads input pixel m, so the im2col index arithmetic (three integer
// divisions per staged row, bounds checks per k-tile) disappears; these are 32 of ResNet-50's 53 convolutions.
// SRC2 (with POINTWISE): the GEMM runs over the concatenated channels of two tensors -- see `in2_buf` in salve_hip.h.
template <int BN, bool POINTWISE, bool SRC2 = false>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_igemm_kernel(ConvArgs p) {
    constexpr int WN = BN / 2;       // wave tile width
    constexpr int NT = WN / 16;      // 16-wide MFMA tiles per wave along n
    constexpr int B_LOADS = BN / 32; // 16-byte chunks of the weight tile per thread
    constexpr int LDC = BN + 8;
    constexpr int STAGE_ELEMS = (BM + BN) * BK;
    constexpr int C_ELEMS = BM * LDC;
    __shared__ __attribute__((aligned(1024))) uint16_t smem[STAGE_ELEMS > C_ELEMS ? STAGE_ELEMS : C_ELEMS];

    int m_tile, n_tile;
    if (!xcd_tile(blockIdx.x, p.m_tiles, p.n_tiles, p.xcd_contig, m_tile, n_tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases need no per-load readfirstlane
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int row_base = tid >> 3;                           // 0..31, plus i * 32
    const int chunk = (tid & 7) ^ ((row_base >> 1) & 7);     // the k-chunk this thread fetches (source-side swizzle)

    // per-thread im2col rows (fixed for the whole K loop)
    int iy0[4], ix0[4];
    const uint16_t* rowp[4];  // POINTWISE: the input pixel's channels (or the zero page, with stride 0)
    const uint16_t* rowp2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int m = m0 + row_base + i * 32;
        const bool valid = m < p.M;
        rowp2[i] = nullptr;
        if (SRC2 && valid) {
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            rowp2[i] = p.in2 + (((long long)b * p.Hi2 + (long long)oy * p.stride2) * p.Wi2 + (long long)ox * p.stride2) * p.Cin2 + chunk * 8;
        }
        if (POINTWISE) {
            rowp[i] = valid ? p.in + (long long)m * p.Cin + chunk * 8 : nullptr;
            iy0[i] = ix0[i] = 0;
        } else {
            const int mm = valid ? m : 0;
            const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            iy0[i] = valid ? oy * p.stride - p.pad : -100000;  // rows beyond M read zeros
            ix0[i] = ox * p.stride - p.pad;
            // the (possibly out-of-image) pixel of tap (0, 0): all 64-bit arithmetic happens here, once.  A k-tile adds one
            // 32-bit element offset (tap shift + channel offset) shared by the thread's four rows, and tests the bounds with
            // two unsigned compares -- no multiplies, no branches.  The vector instructions a k-tile spends on addresses
            // compete with the MFMAs for the SIMD's issue slot: with the full 64-bit index arithmetic per load the 3x3
            // shapes of ResNet-50 ran 6 % slower (batch 512: 887 -> 833 us over the four of them).
            rowp[i] = p.in + (((long long)b * p.Hi + (valid ? iy0[i] : 0)) * p.Wi + ix0[i]) * p.Cin;
        }
    }
    const uint16_t* wrow = p.w + (long long)(n0 + row_base) * p.K + chunk * 8;

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = p.K / BK;

#define ISSUE_TILE(KT, STAGE, E)                                                                                       \
    {                                                                                                                  \
        uint16_t* As_ = smem + (STAGE) * STAGE_ELEMS + wave * 8 * BK;                                                  \
        uint16_t* Bs_ = As_ + BM * BK;                                                                                 \
        const int dy_ = (int8_t)((E) & 0xFF), dx_ = (int8_t)(((E) >> 8) & 0xFF), coff_ = ((E) >> 16) & 0xFFFF;          \
        const int delta_ = (dy_ * p.Wi + dx_) * p.Cin + coff_;                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                \
            const uint16_t* src;                                                                                       \
            if (SRC2 && (KT) >= p.nkt1) {                                                                              \
                src = rowp2[i] ? rowp2[i] + ((KT) - p.nkt1) * BK : p.zeros;                                            \
            } else if (POINTWISE) {                                                                                    \
                src = rowp[i] ? rowp[i] + (KT) * BK : p.zeros;                                                         \
            } else {                                                                                                   \
                const bool ok = (unsigned)(iy0[i] + dy_) < (unsigned)p.Hi && (unsigned)(ix0[i] + dx_) < (unsigned)p.Wi; \
                src = ok ? rowp[i] + delta_ : p.zeros;                                                                 \
            }                                                                                                          \
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(As_ + i * 32 * BK), 16, 0, 0);                \
        }                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; j++)                                                            \
            __builtin_amdgcn_global_load_lds((global_cptr)(wrow + (long long)j * 32 * p.K + (KT) * BK),                \
                                             (lds_ptr)(Bs_ + j * 32 * BK), 16, 0, 0);                                  \
    }

    int32_t e_next = POINTWISE ? 0 : p.ktab[chunk];
    const int frag_row = lane & 15, frag_q = lane >> 4, frag_sw = (frag_row >> 1) & 7;
    for (int kt = 0; kt < nkt; kt++) {
        ISSUE_TILE(kt, 0, e_next);
        {   // table entry of the next tile: a plain load, first used in the next iteration
            const int k1 = kt + 1 < nkt ? kt + 1 : nkt - 1;
            if (!POINTWISE) e_next = p.ktab[k1 * 8 + chunk];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of the tile has landed
        __syncthreads();                                  // ... everyone's
        const uint16_t* As = smem;
        const uint16_t* Bs = As + BM * BK;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ks++) {
            act8 af[4], bfr[NT];
            const int slot = ((ks * 4 + frag_q) ^ frag_sw) * 8;
#pragma unroll
            for (int i = 0; i < 4; i++)
                af[i] = *reinterpret_cast<const act8*>(As + (wr * 64 + i * 16 + frag_row) * BK + slot);
#pragma unroll
            for (int j = 0; j < NT; j++)
                bfr[j] = *reinterpret_cast<const act8*>(Bs + (wc * WN + j * 16 + frag_row) * BK + slot);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < NT; j++)
                    // operands swapped: the accumulator is the TRANSPOSED tile, so a lane owns 4 consecutive output
                    // channels of one output pixel (8 bytes of the NHWC row) instead of 4 pixels of one channel
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // everyone is done reading the tile
    }
#undef ISSUE_TILE

    // ---- epilogue: (residual tile ->) LDS, add bias / residual / ReLU in fp32 on the accumulator's own elements,
    //      round once to fp16, then 16-byte coalesced stores.
    uint16_t* Cs = smem;
    constexpr int CH_PER_ROW = BN / 8;
    constexpr int C_ITERS = (BM * CH_PER_ROW) / CONV_THREADS;
    // (one 64-bit row offset per thread, then a scalar step per iteration: a v_mad_i64_i32 per copy otherwise)
    constexpr int ROW_STEP = CONV_THREADS / CH_PER_ROW;
    const int crow = tid / CH_PER_ROW, cch = tid % CH_PER_ROW;
    const long long coff = (long long)(m0 + crow) * p.Cout + n0 + cch * 8;
    const int cstep = ROW_STEP * p.Cout;
    if (p.res) {
#pragma unroll
        for (int it = 0; it < C_ITERS; it++) {
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (m0 + crow + it * ROW_STEP < p.M) v = *reinterpret_cast<const uint4*>(p.res + coff + it * cstep);
            *reinterpret_cast<uint4*>(Cs + (crow + it * ROW_STEP) * LDC + cch * 8) = v;
        }
        __syncthreads();
    }
    float amax = 0.f;
    float amin = 0.f;
    const float lo = p.relu ? 0.f : -65504.f;
    // (all of the lane's biases in one batch: inside the loop below -- one basic block per tile column because of the residual
    //  branch -- every column's load was followed by its own wait: NT round trips to the L2 per tile, one after the other; round 4)
    float4 bias_j[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) bias_j[j] = *reinterpret_cast<const float4*>(p.bias + n0 + wc * WN + j * 16 + 4 * (lane >> 4));
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int ncol = wc * WN + j * 16 + 4 * (lane >> 4);  // this lane's 4 consecutive channels of tile column j
        const float4 bias = bias_j[j];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int mrow = wr * 64 + i * 16 + (lane & 15);
            uint2* cell = reinterpret_cast<uint2*>(Cs + mrow * LDC + ncol);
            f32x4 v = acc[i][j] + vec4(bias);
            if (p.res) v += vec4(*cell);
            *cell = pack4_lo(amax, amin, v, lo);
        }
    }

    if (!p.relu) amax = fmaxf(amax, -amin);
    report_range(p.status, amax);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < C_ITERS; it++) {
        if (m0 + crow + it * ROW_STEP < p.M)
            *reinterpret_cast<uint4*>(p.out + coff + it * cstep) = *reinterpret_cast<const uint4*>(Cs + (crow + it * ROW_STEP) * LDC + cch * 8);
    }
}

#ifdef SALVE_BUILD_ABLATIONS   // the measured-and-rejected wide-tile kernels d / e / f: tools/probe/build_ablations.sh only
#include "../../tools/probe/ablations/conv_wide.h"
#endif
#include "conv8.h"
#include "stem_pool.h"
#include "expand_chain.h"

// 3x3 / stride 2 / pad 1 max-pool on NHWC fp16, 8 channels (16 bytes) per thread.
__global__ __launch_bounds__(256) void maxpool_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int B,
                                                      int Hi, int Wi, int C, int Ho, int Wo) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c8 = C / 8;
    const long long total = (long long)B * Ho * Wo * c8;
    if (idx >= total) return;
    const int ch = (int)(idx % c8);
    long long t = idx / c8;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float best[8];
#pragma unroll
    for (int k = 0; k < 8; k++) best[k] = -3.0e38f;
    for (int dy = 0; dy < 3; dy++) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= Hi) continue;
        for (int dx = 0; dx < 3; dx++) {
            const int ix = ox * 2 - 1 + dx;
            if (ix < 0 || ix >= Wi) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(in + (((long long)b * Hi + iy) * Wi + ix) * C + ch * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                best[2 * k] = fmaxf(best[2 * k], act_to_f32((uint16_t)(w[k] & 0xFFFFu)));
                best[2 * k + 1] = fmaxf(best[2 * k + 1], act_to_f32((uint16_t)(w[k] >> 16)));
            }
        }
    }
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = (uint32_t)f32_to_act(best[2 * k]) | ((uint32_t)f32_to_act(best[2 * k + 1]) << 16);
    *reinterpret_cast<uint4*>(out + (((long long)b * Ho + oy) * Wo + ox) * C + ch * 8) = uint4{o[0], o[1], o[2], o[3]};
}

// Global average pool over HW positions + fully connected layer (fp32 weights), one block per sample.  A thread owns 8
// consecutive channels (16-byte loads, coalesced across the block) and walks the positions; the per-class partial dot
// products are then reduced across the block.
__global__ __launch_bounds__(256) void avgpool_fc_kernel(const uint16_t* __restrict__ in, int HW, int C,
                                                         const float* __restrict__ fcw, const float* __restrict__ fcb,
                                                         int ncls, float* __restrict__ logits) {
    __shared__ float red[8][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t* x = in + (long long)b * HW * C;
    float part[8];  // ncls <= 8
#pragma unroll
    for (int k = 0; k < 8; k++) part[k] = 0.f;
    const float inv = 1.0f / (float)HW;
    for (int c0 = tid * 8; c0 < C; c0 += 256 * 8) {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; e++) s[e] = 0.f;
        for (int i = 0; i < HW; i++) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + (long long)i * C + c0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s[2 * e] += act_to_f32((uint16_t)(w[e] & 0xFFFFu));
                s[2 * e + 1] += act_to_f32((uint16_t)(w[e] >> 16));
            }
        }
        for (int k = 0; k < ncls; k++) {
            const float* wk = fcw + (long long)k * C + c0;
#pragma unroll
            for (int e = 0; e < 8; e++) part[k] += (s[e] * inv) * wk[e];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        float v = part[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (tid < ncls) logits[(long long)b * ncls + tid] = ((red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3])) + fcb[tid];
}

// ------------------------------------------------------------------------------------------------ fused bottleneck
// One ResNet bottleneck block without down-sampling,  Y = relu(Wc . relu(Wb (*) relu(Wa . X + ba) + bb) + bc + X),
// in ONE kernel: the two intermediate tensors never leave LDS and X is read once (plus a 1-pixel halo) instead of twice.
// For the 56x56 and 28x28 stages the three convolutions are bound by activation traffic (3.3 GB per block and 512
// samples at 56x56); fused, the block moves 1.8 GB.
//
// A workgroup (4 waves) owns a tile of TH x 16 output pixels of one image:
//   GEMM 1  t1[halo pixel][MID]   = relu(X[halo pixel][4 MID] . Wa^T + ba), zero outside the image (that IS the 3x3
//           convolution's zero padding); halo = (TH+2) x 18 pixels, padded to M1 rows; K = 4 MID in k-tiles of 64;
//           operands staged by global_load_lds exactly as in conv_igemm_kernel; result to LDS (fp16, swizzled rows).
//   GEMM 2  t2[pixel][MID] = relu(sum over the 9 taps  t1[pixel + tap][MID] . Wb[tap]^T + bb): the A fragments are read
//           straight out of t1 -- the 16 pixels of a tile row are 16 consecutive halo rows --, Wb streams through LDS.
//   GEMM 3  Y[pixel][4 MID] = relu(t2 . Wc^T + bc + X[pixel]), 128 output channels at a time, epilogue through LDS with
//           16-byte coalesced residual loads and stores as in conv_igemm_kernel.
// MFMA operands are swapped (acc = W-fragment x A-fragment) so that a lane owns 4 consecutive channels of one pixel.
//
// TRANSPOSED TILES (r6).  A 56-pixel image row is 3.5 tiles of 16 columns: with four tile columns the fourth is half empty -- 28 tiles
// per image for 24.5 tiles' worth of pixels, an eighth of the block's three GEMMs multiplying rows that do not exist.  The strip of TH
// columns right of the last whole tile column is therefore covered by tiles of 16 rows x TH columns -- the SAME tile with the roles of
// row and column exchanged (tile pixel (r, c) is image pixel (oy0 + c, ox0 + r)); its halo is the same (TH + 2) x 18 pixels, every LDS
// offset, fragment read and counted wait stays where it is, and only the image addresses (GEMM 1's rows, the output stores) and the
// tap -> halo offset of GEMM 2 exchange their coordinates.  GEMM 2 still walks the taps in the IMAGE's (dy, dx) order, so the sums are
// the same sums in the same order: bit-identical to the three-kernel path.  56 x 56: 21 + 4 = 25 tiles per image instead of 28.
struct BottleneckArgs {
    const uint16_t* x;
    uint16_t* y;
    const uint16_t *wa, *wb, *wc;  // [MID][4 MID], [MID][3][3][MID], [4 MID][MID]  (BatchNorm folded, fp16)
    const float *ba, *bb, *bc;
    const uint16_t* zeros;
    int B, H, W, tiles_x, tiles_y;
    int tiles_t;      // transposed tiles per image (below): 0, or ceil(H / 16) tiles of 16 rows x TH columns for the strip right of tiles_x * 16
    // NEXT form (the last block of layer 1): the FOLLOWING block's first 1x1 convolution rides along as a fourth GEMM on the Y tile
    const uint16_t* wd;   // [MIDN][4 MID] fp16, BatchNorm folded
    const float* bd;
    uint16_t* t1n;        // [B, H, W, MIDN]
    int y_even;           // 1: only the pixels of Y with even row and column are stored (its one other reader samples it at stride 2)
    int32_t* status;
    int xcd_contig;   // 1: every XCD owns a contiguous range of tiles (xcd_linear), so the tiles of one image share an L2
};

constexpr int BN_THREADS = 512;  // 8 waves: two workgroups per CU give 4 waves per SIMD to hide the many short phases

// 16-byte store of an output that nobody reads before the next launch.  (Cache-policy bits on it -- sc1, nt, sc0 sc1, sc1 nt -- were
// timed in round 4 and change nothing; the timing-only builds of this file are tools/probe/ablations/timing_switches.patch.)
__device__ __forceinline__ void store16_stream(uint16_t* p, uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

// PROJ: the first block of layer 1 -- its input has MID channels (not 4 MID) and its shortcut is a 1x1 projection, which rides
//       in GEMM 3 as MID more k: Y = relu([t2 | X] . [Wc | Ws]^T + (bc + bs)), the weight rows K-concatenated as the
//       three-kernel path stores them (conv1x1_with_shortcut), 64 output channels at a time.
// NEXT: the last block of a stage whose successor keeps the resolution in its first convolution (layer 1 -> layer 2: torchvision's v1.5
//       puts the stride on the 3x3) -- that convolution,  t1' = relu(Wd . Y + bd)  with MIDN = 2 MID output channels, is a FOURTH GEMM on
//       the Y tile while each 128-channel chunk of it sits in the staging tile (fp16, exactly the values the separate launch would read
//       back): K = 4 MID in the same ascending k-tiles of 64, so t1' is bit-identical to conv_igemm_kernel's; its accumulators live
//       across the two chunks, Wd streams through the Wc chunk's buffer (and, in the last chunk, through the dead t2).  Y itself is then
//       read from memory only by the next stage's stride-2 projection shortcut: with y_even only a quarter of it is stored.  Per 4096
//       samples that removes the launch that re-read Y (6.6 GB) to write t1' (3.3 GB), and 4.9 GB of Y's stores.
template <int MID, int TH, bool PROJ = false, bool NEXT = false>
__global__ __launch_bounds__(BN_THREADS, 4) void bottleneck_kernel(BottleneckArgs p) {
    static_assert(!NEXT || (!PROJ && MID == 64 && TH == 8), "the NEXT form is written for the plain 64-channel block");
    constexpr int C4 = 4 * MID;
    constexpr int CIN = PROJ ? MID : C4;   // channels of the block's input
    constexpr int HC = 18, HALO = (TH + 2) * HC;
    constexpr int M1 = (HALO + 63) / 64 * 64;   // GEMM-1 rows (halo pixels, zero padded): 192 (TH 8) / 128 (TH 4)
    constexpr int MT1 = M1 / 64;                // GEMM 1: 16-row tiles per wave (4 wave rows x 2 wave columns)
    constexpr int NT1 = MID / 32;               // GEMM 1: 16-channel tiles per wave
    constexpr int MO = TH * 16;                 // output pixels of the tile
    constexpr int RT = TH / 4;                  // GEMM 2/3: 16-pixel row tiles per wave (4 wave rows x 2 wave columns)
    constexpr int NT2 = MID / 32;               // GEMM 2: 16-channel tiles per wave (2 wave columns)
    constexpr int KT_MID = MID / 64;            // k-tiles of 64 in MID
    constexpr int LDC = 128 + 8;
    constexpr int STAGE_TILES = MID == 64 ? 3 : 1;  // GEMM 2: 64-deep Wb tiles per stage (MID 64: a whole kernel row)
    // LDS map (uint16 elements); every phase double-buffers its streamed operand inside the same 72 KB:
    //   GEMM 1   stage s at [s * ST1_E, ...): X tile (M1 x 64) + Wa tile (MID x 64)
    //   t1       [0, T1_E)                           written after GEMM 1's last barrier
    //   GEMM 2   Wb stage s at [T1_E + s * BSB_E, ...)
    //   t2       [0, T2_E)                           written after GEMM 2's last barrier (t1 is dead)
    //   GEMM 3   Wc chunk at [T2_E, ...), output staging behind it
    constexpr int ST1_E = (M1 + MID) * 64;
    constexpr int T1_E = M1 * MID, T2_E = MO * MID;
    constexpr int BSB_E = STAGE_TILES * MID * 64;
    constexpr int BSC_E = 128 * MID, CS_E = MO * LDC;
    constexpr int LDC_P = 64 + 8;                                           // PROJ: 64 output channels per chunk
    constexpr int EC_P = T2_E + MO * 64 + 64 * (MID + CIN) + MO * LDC_P;    // PROJ: t2 | X centre tile | weight chunk | staging
    constexpr int EA = 2 * ST1_E, EB = T1_E + 2 * BSB_E, EC = PROJ ? EC_P : T2_E + BSC_E + CS_E;
    constexpr int SMEM_E = EA > EB ? (EA > EC ? EA : EC) : (EB > EC ? EB : EC);
    __shared__ __attribute__((aligned(1024))) uint16_t smem[SMEM_E];
    uint16_t* T1 = smem;
    uint16_t* T2 = smem;
    uint16_t* BsB = smem + T1_E;
    uint16_t* BsC = smem + T2_E;
    uint16_t* Cs = BsC + BSC_E;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases need no per-load readfirstlane
    const int wr = wave >> 1, wc = wave & 1;   // GEMM 2 / 3
    const int wm = wave & 3, wn = wave >> 2;   // GEMM 1
    const int t = xcd_linear(blockIdx.x, gridDim.x, p.xcd_contig);
    const int n_upright = p.tiles_x * p.tiles_y, per_image = n_upright + p.tiles_t;
    const int b = t / per_image, kt_img = t % per_image;
    const bool TR = kt_img >= n_upright;   // a transposed tile: 16 image rows x TH image columns (workgroup-uniform)
    const int oy0 = TR ? (kt_img - n_upright) * 16 : (kt_img / p.tiles_x) * TH;
    const int ox0 = TR ? p.tiles_x * 16 : (kt_img % p.tiles_x) * 16;
    // tile pixel (r, c) -> image pixel
    auto img_y = [&](int r, int c) { return oy0 + (TR ? c : r); };
    auto img_x = [&](int r, int c) { return ox0 + (TR ? r : c); };
    const uint16_t* ximg = p.x + (long long)b * p.H * p.W * CIN;
    const int frag_row = lane & 15, frag_q = lane >> 4;
    const int row_base = tid >> 3;
    const int chunk = (tid & 7) ^ ((row_base >> 1) & 7);   // source-side swizzle, as in conv_igemm_kernel
    float amax = 0.f;

    // Wb stage `st` -> buffer st & 1 (6 or 8 loads per thread)
#define ISSUE_WB(ST)                                                                                                   \
    {                                                                                                                  \
        _Pragma("unroll") for (int q = 0; q < STAGE_TILES; q++) {                                                      \
            const int ktile_ = (ST) * STAGE_TILES + q;                                                                 \
            _Pragma("unroll") for (int j = 0; j < MID / 64; j++)                                                       \
                __builtin_amdgcn_global_load_lds(                                                                      \
                    (global_cptr)(p.wb + (long long)(row_base + 64 * j) * (9 * MID) + ktile_ * 64 + chunk * 8),        \
                    (lds_ptr)(BsB + ((ST) & 1) * BSB_E + q * MID * 64 + (wave * 8 + 64 * j) * 64), 16, 0, 0);          \
        }                                                                                                              \
    }
    constexpr int WB_LOADS = STAGE_TILES * (MID / 64);

    // The residual of GEMM 3 is the block's own input at the output pixels -- a subset of the halo tile GEMM 1 streams through
    // LDS.  Each thread copies the 8-byte pieces its GEMM-3 accumulators will need (pixel rows wr * RT + i, the 16 channels per
    // accumulator tile of wave column wc: k-tile kt serves the waves with wc == (kt & 1), chunk nc = kt >> 1) out of the
    // stage buffers into registers while the tile is there: 32 VGPRs instead of a second trip to memory for 64 KB per tile
    // (round 3, PMC: the re-read came from beyond the L2 and was 30 % of the kernel's HBM traffic).
    uint2 resid[PROJ ? 1 : C4 / 128][RT][4];
    int res_off[RT];   // element offset of the pixel's row in a stage buffer, and its swizzle
    int res_swz[RT];
#pragma unroll
    for (int i = 0; i < RT; i++) {
        const int m = (wr * RT + i) * 16 + frag_row;
        const int h = ((m >> 4) + 1) * HC + (m & 15) + 1;
        res_off[i] = h * 64 + 4 * (frag_q & 1);
        res_swz[i] = (h >> 1) & 7;
    }

    // ------------------------------------------------------------------ GEMM 1: t1 = relu(Xhalo . Wa^T + ba)
    {
        const uint16_t* rowp[M1 / 64];
#pragma unroll
        for (int i = 0; i < M1 / 64; i++) {
            const int h = row_base + 64 * i;
            const int hy = img_y(h / HC - 1, h % HC - 1), hx = img_x(h / HC - 1, h % HC - 1);
            const bool ok = h < HALO && hy >= 0 && hy < p.H && hx >= 0 && hx < p.W;
            rowp[i] = ok ? ximg + ((long long)hy * p.W + hx) * CIN + chunk * 8 : nullptr;
        }
        const uint16_t* wrow = p.wa + (long long)row_base * CIN + chunk * 8;
        constexpr int A_LOADS = M1 / 64 + MID / 64;
#define ISSUE_A(KT)                                                                                                    \
    {                                                                                                                  \
        uint16_t* As_ = smem + ((KT) & 1) * ST1_E;                                                                     \
        _Pragma("unroll") for (int i = 0; i < M1 / 64; i++) {                                                          \
            const uint16_t* src = rowp[i] ? rowp[i] + (KT) * 64 : p.zeros;                                             \
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(As_ + (wave * 8 + 64 * i) * 64), 16, 0, 0);   \
        }                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < MID / 64; j++)                                                           \
            __builtin_amdgcn_global_load_lds((global_cptr)(wrow + (long long)j * 64 * CIN + (KT) * 64),                \
                                             (lds_ptr)(As_ + M1 * 64 + (wave * 8 + 64 * j) * 64), 16, 0, 0);           \
    }
        f32x4 acc[MT1][NT1];
#pragma unroll
        for (int i = 0; i < MT1; i++)
#pragma unroll
            for (int j = 0; j < NT1; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NK1 = CIN / 64;
        ISSUE_A(0);
        // (Round 3: touching the later k-tiles' lines up front -- one 4-byte LDS-DMA lane per 128-byte line into scratch LDS, so
        // that the stages after the first hit the L2 -- measured +1 % on the forward: the phases are not waiting for those lines.)
#pragma unroll
        for (int kt = 0; kt < NK1; kt++) {
            // the next tile goes into the other buffer (last read one iteration ago, behind a barrier) and stays in
            // flight under this tile's MFMAs: counted wait, raw barriers (a __syncthreads() would drain it)
            if (kt + 1 < NK1) {
                ISSUE_A(kt + 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LOADS) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            const uint16_t* As = smem + (kt & 1) * ST1_E;
            const uint16_t* BsA = As + M1 * 64;
            if constexpr (!PROJ) {
                if (wc == (kt & 1)) {
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            resid[kt >> 1][i][j] = lds_read8(As + res_off[i] + (((2 * j + (frag_q >> 1)) ^ res_swz[i]) << 3));   // (typed so that no vmcnt(0) lands in front of it: the next k-tile is in flight)
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const int slot = ((ks * 4 + frag_q) ^ ((frag_row >> 1) & 7)) * 8;
                act8 af[MT1], bfr[NT1];
#pragma unroll
                for (int i = 0; i < MT1; i++)
                    af[i] = *reinterpret_cast<const act8*>(As + ((wm * MT1 + i) * 16 + frag_row) * 64 + slot);
#pragma unroll
                for (int j = 0; j < NT1; j++)
                    bfr[j] = *reinterpret_cast<const act8*>(BsA + ((wn * NT1 + j) * 16 + frag_row) * 64 + slot);
#pragma unroll
                for (int i = 0; i < MT1; i++)
#pragma unroll
                    for (int j = 0; j < NT1; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");  // everyone is done reading this buffer
        }
#undef ISSUE_A
        ISSUE_WB(0);  // first Wb stage: lands while t1 is written (its buffer is behind t1)
        // t1 rows: MID halves = MID / 8 chunks of 16 bytes, chunk q of row h stored at slot q ^ swz(h)
#pragma unroll
        for (int i = 0; i < MT1; i++) {
            const int h = (wm * MT1 + i) * 16 + frag_row;
            const int hy = img_y(h / HC - 1, h % HC - 1), hx = img_x(h / HC - 1, h % HC - 1);
            const bool inside = h < HALO && hy >= 0 && hy < p.H && hx >= 0 && hx < p.W;
            const uint32_t keep = inside ? 0xFFFFFFFFu : 0u;   // (a mask, not a select around the conversions: hipcc made branches of those)
            const int swz = MID == 64 ? ((h >> 1) & 7) : (h & 15);
#pragma unroll
            for (int j = 0; j < NT1; j++) {
                const int c0 = (wn * NT1 + j) * 16 + 4 * frag_q;
                const float4 bias = *reinterpret_cast<const float4*>(p.ba + c0);
                uint2 o = pack4<true>(amax, acc[i][j] + vec4(bias));
                o.x &= keep; o.y &= keep;
                *reinterpret_cast<uint2*>(T1 + h * MID + (((c0 >> 3) ^ swz) << 3) + (c0 & 7)) = o;
            }
        }
    }

    // ------------------------------------------------------------------ GEMM 2: t2 = relu(3x3(t1) . Wb + bb)
    {
        f32x4 acc[RT][NT2];
#pragma unroll
        for (int i = 0; i < RT; i++)
#pragma unroll
            for (int j = 0; j < NT2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int N_STAGES = 9 * KT_MID / STAGE_TILES;
        for (int st = 0; st < N_STAGES; st++) {
            if (st + 1 < N_STAGES) {
                ISSUE_WB(st + 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WB_LOADS) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (first stage: this wave's t1 stores)
            asm volatile("s_barrier" ::: "memory");
            const uint16_t* Bst = BsB + (st & 1) * BSB_E;
#pragma unroll
            for (int q = 0; q < STAGE_TILES; q++) {
                const int ktile = st * STAGE_TILES + q;
                const int tap = ktile / KT_MID, kh = ktile % KT_MID;  // kh: which 64 channels of t1
                const int dy = tap / 3, dx = tap % 3;
                const int ro = TR ? dx : dy, co = TR ? dy : dx;   // the tap's offset in TILE rows / columns (transposed tiles: exchanged)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    act8 af[RT], bfr[NT2];
#pragma unroll
                    for (int i = 0; i < RT; i++) {
                        const int h = (wr * RT + i + ro) * HC + co + frag_row;
                        const int swz = MID == 64 ? ((h >> 1) & 7) : (h & 15);
                        af[i] = *reinterpret_cast<const act8*>(T1 + h * MID + (((kh * 8 + ks * 4 + frag_q) ^ swz) << 3));
                    }
                    const int slot = ((ks * 4 + frag_q) ^ ((frag_row >> 1) & 7)) * 8;
#pragma unroll
                    for (int j = 0; j < NT2; j++)
                        bfr[j] = *reinterpret_cast<const act8*>(Bst + q * MID * 64 + ((wc * NT2 + j) * 16 + frag_row) * 64 + slot);
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int j = 0; j < NT2; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const int m = (wr * RT + i) * 16 + frag_row;  // output pixel of the tile, row-major
            const int swz = MID == 64 ? ((m >> 1) & 7) : (m & 15);
#pragma unroll
            for (int j = 0; j < NT2; j++) {
                const int c0 = (wc * NT2 + j) * 16 + 4 * frag_q;
                const float4 bias = *reinterpret_cast<const float4*>(p.bb + c0);
                *reinterpret_cast<uint2*>(T2 + m * MID + (((c0 >> 3) ^ swz) << 3) + (c0 & 7)) =
                    pack4<true>(amax, acc[i][j] + vec4(bias));
            }
        }
    }
#undef ISSUE_WB

    // ------------------------------------------------------------------ GEMM 3 (PROJ): Y = relu([t2 | X] . [Wc | Ws]^T + b)
    if constexpr (PROJ) {
        static_assert(!PROJ || MID == 64, "the projection variant is written for 64 mid channels");
        constexpr int KP = MID + CIN;               // 128: k-tile 0 = t2, k-tile 1 = the block's input at the output pixels
        uint16_t* Xc = smem + T2_E;                 // [MO][64], staged like a GEMM-1 tile (chunk q of row r at q ^ ((r >> 1) & 7))
        uint16_t* Wp = Xc + MO * 64;                // weight chunk: two tiles of 64 rows x 64 k
        uint16_t* Cp = Wp + 64 * KP;                // output staging [MO][LDC_P]
        constexpr int CH_PER_ROW = 64 / 8;
        constexpr int C_ITERS = (MO * CH_PER_ROW) / BN_THREADS;
#pragma unroll
        for (int i = 0; i < MO / 64; i++) {         // X centre tile (L2: GEMM 1 has just read it), lands with the first weights
            const int r = row_base + 64 * i;
            const int oy = img_y(r >> 4, r & 15), ox = img_x(r >> 4, r & 15);
            const uint16_t* src = (ox < p.W && oy < p.H) ? ximg + ((long long)oy * p.W + ox) * CIN + chunk * 8 : p.zeros;
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(Xc + (wave * 8 + 64 * i) * 64), 16, 0, 0);
        }
        for (int nc = 0; nc < C4 / 64; nc++) {
#pragma unroll
            for (int q = 0; q < KP / 64; q++)
                __builtin_amdgcn_global_load_lds((global_cptr)(p.wc + (long long)(nc * 64 + row_base) * KP + q * 64 + chunk * 8),
                                                 (lds_ptr)(Wp + q * 64 * 64 + (wave * 8) * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // (first chunk: also orders the t2 stores)
            f32x4 acc[RT][2];
#pragma unroll
            for (int i = 0; i < RT; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < KP / 64; q++)
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    act8 af[RT], bfr[2];
                    const uint16_t* A = q == 0 ? T2 : Xc;
#pragma unroll
                    for (int i = 0; i < RT; i++) {
                        const int m = (wr * RT + i) * 16 + frag_row;
                        af[i] = *reinterpret_cast<const act8*>(A + m * 64 + (((ks * 4 + frag_q) ^ ((m >> 1) & 7)) << 3));
                    }
                    const int slot = ((ks * 4 + frag_q) ^ ((frag_row >> 1) & 7)) * 8;
#pragma unroll
                    for (int j = 0; j < 2; j++)
                        bfr[j] = *reinterpret_cast<const act8*>(Wp + q * 64 * 64 + ((wc * 2 + j) * 16 + frag_row) * 64 + slot);
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int ncol = (wc * 2 + j) * 16 + 4 * frag_q;
                const float4 bias = *reinterpret_cast<const float4*>(p.bc + nc * 64 + ncol);
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    const int m = (wr * RT + i) * 16 + frag_row;
                    *reinterpret_cast<uint2*>(Cp + m * LDC_P + ncol) =
                        pack4<true>(amax, acc[i][j] + vec4(bias));
                }
            }
            __syncthreads();
            uint16_t* yimg = p.y + (long long)b * p.H * p.W * C4;
#pragma unroll
            for (int it = 0; it < C_ITERS; it++) {
                const int id = tid + it * BN_THREADS;
                const int m = id / CH_PER_ROW, ch = id % CH_PER_ROW;
                const int oy = img_y(m >> 4, m & 15), ox = img_x(m >> 4, m & 15);
                if (ox < p.W && oy < p.H)
                    store16_stream(yimg + ((long long)oy * p.W + ox) * C4 + nc * 64 + ch * 8, *reinterpret_cast<const uint4*>(Cp + m * LDC_P + ch * 8));
            }
            __syncthreads();  // staging and weight chunk are reused by the next chunk
        }
    } else
    // ------------------------------------------------------------------ GEMM 3: Y = relu(t2 . Wc^T + bc + X)   (+ NEXT: GEMM 4 on Y)
    {
        constexpr int CH_PER_ROW = 128 / 8;
        constexpr int C_ITERS = (MO * CH_PER_ROW) / BN_THREADS;
        constexpr int MIDN = 2 * MID;                        // NEXT: output channels of the following block's first convolution (128)
        f32x4 acc4[NEXT ? RT : 1][NEXT ? MIDN / 32 : 1];     // NEXT: t1' accumulators, 16 pixels x 16 channels each, alive across both chunks
        // NEXT: Wd k-tile KT (MIDN rows x 64 k of the [MIDN][4 MID] matrix) -> DST, laid out like a Wc tile (128 rows x 128 bytes, source-side swizzle)
#define ISSUE_WD(KT, DST)                                                                                              \
    {                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < MIDN / 64; j++)                                                          \
            __builtin_amdgcn_global_load_lds((global_cptr)(p.wd + (long long)(row_base + 64 * j) * C4 + (KT) * 64 + chunk * 8), \
                                             (lds_ptr)((DST) + (wave * 8 + 64 * j) * 64), 16, 0, 0);                   \
    }
        // NEXT: one k-tile of GEMM 4: A = 64 channels of the Y chunk in the staging tile (rows of LDC halves), B = a Wd tile
#define GEMM4_KTILE(ACOL, BT)                                                                                          \
    {                                                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                                             \
            act8 af[RT], bfr[MIDN / 32];                                                                               \
            _Pragma("unroll") for (int i = 0; i < RT; i++)                                                             \
                af[i] = *reinterpret_cast<const act8*>(Cs + ((wr * RT + i) * 16 + frag_row) * LDC + (ACOL) + ks * 32 + frag_q * 8); \
            const int slot = ((ks * 4 + frag_q) ^ ((frag_row >> 1) & 7)) * 8;                                          \
            _Pragma("unroll") for (int j = 0; j < MIDN / 32; j++)                                                      \
                bfr[j] = *reinterpret_cast<const act8*>((BT) + ((wc * (MIDN / 32) + j) * 16 + frag_row) * 64 + slot);  \
            _Pragma("unroll") for (int i = 0; i < RT; i++)                                                             \
                _Pragma("unroll") for (int j = 0; j < MIDN / 32; j++)                                                  \
                    acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc4[i][j], 0, 0, 0);          \
        }                                                                                                              \
    }
#pragma unroll
        for (int nc = 0; nc < C4 / 128; nc++) {
            // Wc chunk: 128 output channels x MID, as KT_MID tiles of 128 rows x 64
#pragma unroll
            for (int q = 0; q < KT_MID; q++)
#pragma unroll
                for (int j = 0; j < 2; j++)
                    __builtin_amdgcn_global_load_lds((global_cptr)(p.wc + (long long)(nc * 128 + row_base + 64 * j) * MID + q * 64 + chunk * 8),
                                                     (lds_ptr)(BsC + q * 128 * 64 + (wave * 8 + 64 * j) * 64), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // (first chunk: also orders the t2 stores)
            // (NEXT: the chunk in two passes of 64 channels -- 16 accumulator registers next to the 32 of GEMM 4 and the residual's; in one
            //  pass the kernel needs 50 registers more than the 128 that four waves per SIMD leave, and a scratch reload is a vmcnt load
            //  queued behind the LDS-DMA stages)
            constexpr int NH = NEXT ? 2 : 1, JN = 4 / NH;
#pragma unroll
            for (int hf = 0; hf < NH; hf++) {
                f32x4 acc[RT][JN];
#pragma unroll
                for (int i = 0; i < RT; i++)
#pragma unroll
                    for (int j = 0; j < JN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < KT_MID; q++)
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        act8 af[RT], bfr[JN];
#pragma unroll
                        for (int i = 0; i < RT; i++) {
                            const int m = (wr * RT + i) * 16 + frag_row;
                            const int swz = MID == 64 ? ((m >> 1) & 7) : (m & 15);
                            af[i] = *reinterpret_cast<const act8*>(T2 + m * MID + (((q * 8 + ks * 4 + frag_q) ^ swz) << 3));
                        }
                        const int slot = ((ks * 4 + frag_q) ^ ((frag_row >> 1) & 7)) * 8;
#pragma unroll
                        for (int j = 0; j < JN; j++)
                            bfr[j] = *reinterpret_cast<const act8*>(BsC + q * 128 * 64 + ((wc * 4 + hf * JN + j) * 16 + frag_row) * 64 + slot);
#pragma unroll
                        for (int i = 0; i < RT; i++)
#pragma unroll
                            for (int j = 0; j < JN; j++)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                    }
                if constexpr (NEXT) {
                    if (hf == NH - 1) {
                        // everyone is done with the Wc chunk (and, in the last chunk, with t2): the first Wd k-tile of this chunk lands in the Wc
                        // buffer while the epilogue below runs; the last chunk brings both of its k-tiles, the second into the dead t2
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        asm volatile("s_barrier" ::: "memory");
                        ISSUE_WD(2 * nc, BsC);
                        if (nc == C4 / 128 - 1) ISSUE_WD(2 * nc + 1, T2);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < JN; jj++) {
                    const int j = hf * JN + jj;
                    const int ncol = (wc * 4 + j) * 16 + 4 * frag_q;
                    const float4 bias = *reinterpret_cast<const float4*>(p.bc + nc * 128 + ncol);
#pragma unroll
                    for (int i = 0; i < RT; i++) {
                        const int m = (wr * RT + i) * 16 + frag_row;
                        const uint2 r = resid[nc][i][j];   // (a pixel beyond the image's right edge holds zeros: its halo row was the zero page)
                        const uint2 o = pack4<true>(amax, acc[i][jj] + vec4(bias) + vec4(r));
                        if constexpr (NEXT) lds_write8(Cs + m * LDC + ncol, o);   // (typed: no compiler vmcnt(0) in front of it -- the Wd tile is in flight)
                        else *reinterpret_cast<uint2*>(Cs + m * LDC + ncol) = o;
                    }
                }
            }
            if constexpr (NEXT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the Wd tile(s)
            __syncthreads();
            uint16_t* yimg = p.y + (long long)b * p.H * p.W * C4;
            int tl = tid;
            if constexpr (NEXT) asm volatile("" : "+v"(tl));   // (opaque: the store addresses are computed HERE, not hoisted above GEMM 1 and kept in registers -- or scratch -- across the kernel)
#pragma unroll
            for (int it = 0; it < C_ITERS; it++) {
                const int id = tl + it * BN_THREADS;
                const int m = id / CH_PER_ROW, ch = id % CH_PER_ROW;
                const int oy = img_y(m >> 4, m & 15), ox = img_x(m >> 4, m & 15);
                bool keep = ox < p.W && oy < p.H;
                if constexpr (NEXT) keep = keep && (!p.y_even || (((oy | ox) & 1) == 0));
                if (keep)
                    store16_stream(yimg + ((long long)oy * p.W + ox) * C4 + nc * 128 + ch * 8, *reinterpret_cast<const uint4*>(Cs + m * LDC + ch * 8));
            }
            if constexpr (NEXT) {
                if (nc == 0) {   // (zeroed here, not in front of the chunk loop: 32 registers that are not live through the first chunk's GEMM 3)
#pragma unroll
                    for (int i = 0; i < RT; i++)
#pragma unroll
                        for (int j = 0; j < MIDN / 32; j++) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                GEMM4_KTILE(0, BsC);
                if (nc == C4 / 128 - 1) {
                    GEMM4_KTILE(64, T2);
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");   // everyone is done with the first Wd k-tile
                    ISSUE_WD(2 * nc + 1, BsC);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                    GEMM4_KTILE(64, BsC);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __syncthreads();  // staging and Wc tile are reused by the next chunk
        }
        if constexpr (NEXT) {
            // t1' = relu(acc4 + bd): through the staging tile (free now), then 16-byte coalesced stores of whole 256-byte pixel rows
#pragma unroll
            for (int j = 0; j < MIDN / 32; j++) {
                const int ncol = (wc * (MIDN / 32) + j) * 16 + 4 * frag_q;
                const float4 bias = *reinterpret_cast<const float4*>(p.bd + ncol);
#pragma unroll
                for (int i = 0; i < RT; i++) {
                    const int m = (wr * RT + i) * 16 + frag_row;
                    *reinterpret_cast<uint2*>(Cs + m * LDC + ncol) = pack4<true>(amax, acc4[i][j] + vec4(bias));
                }
            }
            __syncthreads();
            uint16_t* timg = p.t1n + (long long)b * p.H * p.W * MIDN;
            constexpr int CPR = MIDN / 8;
            int tl = tid;
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int it = 0; it < (MO * CPR) / BN_THREADS; it++) {
                const int id = tl + it * BN_THREADS;
                const int m = id / CPR, ch = id % CPR;
                const int oy = img_y(m >> 4, m & 15), ox = img_x(m >> 4, m & 15);
                if (ox < p.W && oy < p.H)
                    store16_stream(timg + ((long long)oy * p.W + ox) * MIDN + ch * 8, *reinterpret_cast<const uint4*>(Cs + m * LDC + ch * 8));
            }
        }
#undef ISSUE_WD
#undef GEMM4_KTILE
    }
    report_range(p.status, amax);
}

struct ResnetHandle {
    std::vector<salve_resnet_op_t> ops;
    uint16_t* d_weights = nullptr;
    float* d_params = nullptr;
    int32_t* d_ktab = nullptr;
    uint16_t* d_zeros = nullptr;
    size_t max_act_elems = 0;  // per sample, elements of the largest activation buffer
    int n_bufs = 0;
    int num_layers = 0, in_channels = 0, ncls = 0;
    std::vector<int> fused;  // per op: 1 = this op and the next two form a bottleneck block run by bottleneck_kernel
    std::vector<int> wide;   // per op: 0 = conv_igemm_kernel, else a conv_wide_kernel configuration (WIDE_*)
    bool wide_auto = false;  // the 8-phase kernel only for launches that fill the chip at least twice over (256 x 256 tiles)
    std::vector<int> stem;   // per op: 1 = this 7x7 / 2 convolution and the max-pool behind it run as stem_pool_kernel
    int xcd_contig = 1;      // SALVE_XCD_CONTIG=0: the round-1 mapping (m-tiles dealt round-robin to the XCDs), for A/B runs
    std::vector<int> chain;  // per op: 1 = this expand convolution (+ residual) runs as expand_chain_kernel, 2 = ... together with
                             //         the next op, the following block's first 1x1 convolution (expand_chain.h)
    int n_cus = 256;
    int chain_split = 1;                  // SALVE_RESNET_CHAIN_NO_SPLIT: the 8-wave form for the 256-channel shapes too
    int chain_dbg = 0, chain_waves = 8;   // chain_dbg: ablation build only (timing-only switches of expand_chain_kernel); SALVE_RESNET_CHAIN_16_WAVES
    std::vector<int> y_even;              // per op: 1 = the chained expand convolution stores only Y's even pixels (expand_chain.h: y_even_w)
    std::vector<int> next;                // per op: 1 = this fused block also computes the op three further on (the next block's first 1x1 convolution:
                                          //         bottleneck_kernel's NEXT form), 2 = ... and stores only the even pixels of its own output
    int no_transposed_tiles = 0;          // SALVE_RESNET_NO_TRANSPOSED_TILES: the fused blocks' fourth, half-empty tile column instead (bit-identity tests, A/B)
};

// expand_chain_kernel shapes: (MID, MIDN) of the chained form, MID of the expand-only form
static bool chain_shape(int mid, int midn) { return (mid == 128 && (midn == 128 || midn == 256)) || (mid == 256 && midn == 256); }
static bool expand_shape(int mid) { return mid == 128 || mid == 256; }   // (512: measured 0.80 ms against 0.71 ms for conv8_kernel at batch 4096: compute-bound there)

// SALVE_CONV_WIDE (read when a handle is created): unset = the 8-phase 256 x 256 kernel (conv8.h) on the compute-bound shapes
// that fit it, 0 = conv_igemm_kernel everywhere, 8 = the 8-phase kernel wherever it fits.  d | e | f select the wide-tile /
// split-role kernels of conv_wide.h, which were measured per shape on MI355X and rejected: they exist
// only in the ablation build (-DSALVE_BUILD_ABLATIONS, tools/probe/build_ablations.sh) -- the product library treats them as 0.
// All of them are bit-identical to conv_igemm_kernel (same k order, same fp32 accumulation).
enum { WIDE_OFF = 0, WIDE_256_K64_S2 = 4, WIDE_128_K64_S1 = 5, WIDE_PC_128 = 6, WIDE_8PHASE = 7, WIDE_AUTO = 8 };

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static int choose_wide(const salve_resnet_op_t& o, int force) {
    if (o.op != SALVE_OP_CONV || force == WIDE_OFF) return WIDE_OFF;
    const bool pointwise = o.KH == 1 && o.KW == 1 && o.stride == 1 && o.pad == 0;
    const int K = o.KH * o.KW * o.Cin + (o.in2_buf != SALVE_NO_BUF ? o.Cin2 : 0);
    if (!pointwise && (!is_pow2(o.Cin) || o.Cin < 64)) return WIDE_OFF;   // the stem keeps its table-driven gather
    if (K < 128 || K % 64 != 0) return WIDE_OFF;
    if (force == WIDE_8PHASE || force == WIDE_AUTO) {
        const bool fits = o.Cout % 256 == 0 && is_pow2(o.Cin) && o.Cin >= 64 && (o.in2_buf == SALVE_NO_BUF || o.Cin2 % 64 == 0);
        // by default only where the convolution is compute-bound (measured per shape): k >= 512
        // K = 384 (layer 2's projection pair) measured the same on either kernel (r3)
        return (fits && (force == WIDE_8PHASE || K >= 512)) ? WIDE_8PHASE : WIDE_OFF;
    }
    if (force == WIDE_256_K64_S2) return o.Cout % 256 == 0 ? force : WIDE_OFF;
    return o.Cout % 128 == 0 ? force : WIDE_OFF;
}

bool check_op(const salve_resnet_op_t& o) {
    if (o.op == SALVE_OP_CONV) {
        if (o.Cout % 64 != 0) return salve_fail("conv: Cout must be a multiple of 64");
        if (o.Cin % 8 != 0) return salve_fail("conv: Cin must be a multiple of 8");
        if ((o.KH * o.KW * o.Cin) % BK != 0) return salve_fail("conv: KH*KW*Cin must be a multiple of 64");
        if (o.in2_buf != SALVE_NO_BUF) {
            if (o.KH != 1 || o.KW != 1 || o.stride != 1 || o.pad != 0 || o.res_buf != SALVE_NO_BUF)
                return salve_fail("conv: a second source needs a 1x1 / stride 1 / pad 0 convolution without residual");
            if (o.Cin2 <= 0 || o.Cin2 % BK != 0 || o.stride2 < 1 || (o.Hi2 - 1) / o.stride2 + 1 != o.Ho || (o.Wi2 - 1) / o.stride2 + 1 != o.Wo)
                return salve_fail("conv: bad second-source geometry");
        }
    } else if (o.op == SALVE_OP_MAXPOOL) {
        if (o.Cin % 8 != 0) return salve_fail("maxpool: C must be a multiple of 8");
    } else if (o.op == SALVE_OP_AVGPOOL_FC) {
        if (o.Cout < 1 || o.Cout > 8) return salve_fail("fc: 1..8 classes supported");
        if (o.Cin % 8 != 0) return salve_fail("fc: C must be a multiple of 8");
    } else {
        return salve_fail("unknown op");
    }
    return true;
}

}  // namespace

extern "C" {



void* salve_resnet_create(int32_t num_layers, int32_t in_channels, const salve_resnet_op_t* ops, int32_t n_ops,
                          const void* weights_f16, size_t weights_bytes, const float* params_f32, size_t params_bytes,
                          const int32_t* ktab, size_t ktab_entries, int32_t flags) {
    if (!ops || n_ops <= 0 || !weights_f16 || !params_f32 || !ktab) {
        salve_fail("salve_resnet_create: null argument");
        return nullptr;
    }
    if (flags & ~SALVE_RESNET_ALL_FLAGS) {
        salve_fail("salve_resnet_create: unknown bit in flags (SALVE_RESNET_*): a caller written for another ABI version");
        return nullptr;
    }
    ResnetHandle* h = new ResnetHandle();
    h->num_layers = num_layers;
    h->in_channels = in_channels;
    for (int i = 0; i < n_ops; i++) {
        if (!check_op(ops[i])) { delete h; return nullptr; }
        h->ops.push_back(ops[i]);
        const salve_resnet_op_t& o = ops[i];
        if (o.op != SALVE_OP_AVGPOOL_FC) {
            const size_t e = (size_t)o.Ho * o.Wo * o.Cout;
            if (e > h->max_act_elems) h->max_act_elems = e;
            if (o.out_buf + 1 > h->n_bufs) h->n_bufs = o.out_buf + 1;
        } else {
            h->ncls = o.Cout;
        }
    }
    h->wide.assign(h->ops.size(), 0);
    {
        // default: the 8-phase kernel on the compute-bound shapes that fit it, in launches large enough for it
        int mode = (flags & SALVE_RESNET_CONV_IGEMM_ONLY) ? WIDE_OFF : ((flags & SALVE_RESNET_CONV8_WHEREVER) ? WIDE_8PHASE : WIDE_AUTO);
#ifdef SALVE_BUILD_ABLATIONS   // development build only: the rejected wide-tile kernels d / e / f of tools/probe/ablations/conv_wide.h
        if (const char* e = getenv("SALVE_CONV_WIDE")) mode = e[0] == 'd' ? WIDE_256_K64_S2 : (e[0] == 'e' ? WIDE_128_K64_S1 : (e[0] == 'f' ? WIDE_PC_128 : mode));
#endif
        h->wide_auto = mode == WIDE_AUTO;
        for (size_t i = 0; i < h->ops.size(); i++) h->wide[i] = choose_wide(h->ops[i], mode);
    }
    h->xcd_contig = (flags & SALVE_RESNET_ROUND_ROBIN_TILES) ? 0 : 1;
    h->no_transposed_tiles = (flags & SALVE_RESNET_NO_TRANSPOSED_TILES) ? 1 : 0;
    h->stem.assign(h->ops.size(), 0);
    {
        const bool enable = !(flags & SALVE_RESNET_NO_STEM_FUSE);   // else: the implicit-GEMM stem and the separate max-pool
        for (size_t i = 0; enable && i + 1 < h->ops.size(); i++) {
            const salve_resnet_op_t &a = h->ops[i], &b = h->ops[i + 1];
            if (a.op != SALVE_OP_CONV || b.op != SALVE_OP_MAXPOOL) continue;
            bool k_order = a.Cin == 8;   // 16 / 24 input channels: the fused kernel needs K ordered (group of 8 channels, kh, kw, channel): read it off the k table
            if ((a.Cin == 16 || a.Cin == 24) && a.KH == 7 && a.KW == 8 && (size_t)a.ktab_off + (size_t)(a.Cin / 8) * 7 * 8 <= ktab_entries) {
                k_order = true;
                for (int q = 0; q < (a.Cin / 8) * 7 * 8 && k_order; q++) {
                    const int g = q / 56, kh = (q % 56) / 8, kw = q % 8;
                    k_order = ktab[a.ktab_off + q] == ((kh & 0xFF) | ((kw & 0xFF) << 8) | ((8 * g) << 16));
                }
            }
            const bool shape = a.KH == 7 && a.KW == 8 && a.stride == 2 && a.pad == 3 && (a.Cin == 8 || a.Cin == 16 || a.Cin == 24) && k_order && a.Cout == 64 && a.relu &&
                               a.res_buf == SALVE_NO_BUF && a.in2_buf == SALVE_NO_BUF && b.in_buf == a.out_buf && b.Cin == 64 &&
                               a.Hi % 4 == 0 && a.Wi % 4 == 0 && a.Wi == STEM_W && (a.Hi / 4) % STEM_R == 0 &&
                               a.Ho == a.Hi / 2 && a.Wo == a.Wi / 2 && b.Ho == a.Hi / 4 && b.Wo == a.Wi / 4;
            bool dead = shape;   // nobody else may read the un-pooled convolution output
            for (size_t k = i + 2; dead && k < h->ops.size(); k++) {
                const salve_resnet_op_t& o = h->ops[k];
                if (o.in_buf == a.out_buf || (o.op == SALVE_OP_CONV && (o.res_buf == a.out_buf || o.in2_buf == a.out_buf))) dead = false;
                if (o.out_buf == a.out_buf) break;
            }
            if (dead) h->stem[i] = 1;
        }
    }
    h->fused.assign(h->ops.size(), 0);
    {
        const bool enable = !(flags & SALVE_RESNET_NO_BLOCK_FUSE);
        for (size_t i = 0; enable && i + 2 < h->ops.size(); i++) {
            const salve_resnet_op_t &a = h->ops[i], &b = h->ops[i + 1], &c = h->ops[i + 2];
            if (a.op != SALVE_OP_CONV || b.op != SALVE_OP_CONV || c.op != SALVE_OP_CONV) continue;
            const int mid = a.Cout;
            const bool shapes = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.relu && a.res_buf == SALVE_NO_BUF && a.Cin == 4 * mid &&
                                b.KH == 3 && b.KW == 3 && b.stride == 1 && b.pad == 1 && b.relu && b.res_buf == SALVE_NO_BUF && b.Cin == mid && b.Cout == mid &&
                                b.in_buf == a.out_buf && a.in2_buf == SALVE_NO_BUF && c.in2_buf == SALVE_NO_BUF && c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad == 0 && c.relu && c.Cin == mid &&
                                c.Cout == 4 * mid && c.in_buf == b.out_buf && c.res_buf == a.in_buf && c.out_buf != a.in_buf && a.in_buf >= 0 &&
                                a.Hi == c.Ho && a.Wi == c.Wo && b.Hi == a.Hi && b.Ho == a.Hi;
            // measured at batch 512: the 64-channel blocks (56 x 56) gain 15 % fused; the 128-channel blocks (28 x 28, 4 x 16
            // tiles) come out even, so they stay on the three-kernel path
            // ... or the first block of layer 1: input of `mid` channels, the projection shortcut in the last convolution's k
            const bool proj = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.relu && a.res_buf == SALVE_NO_BUF && a.in2_buf == SALVE_NO_BUF &&
                              a.Cin == mid && b.KH == 3 && b.KW == 3 && b.stride == 1 && b.pad == 1 && b.relu && b.res_buf == SALVE_NO_BUF &&
                              b.in2_buf == SALVE_NO_BUF && b.Cin == mid && b.Cout == mid && b.in_buf == a.out_buf && c.KH == 1 && c.KW == 1 && c.stride == 1 &&
                              c.pad == 0 && c.relu && c.res_buf == SALVE_NO_BUF && c.Cin == mid && c.Cout == 4 * mid && c.in_buf == b.out_buf &&
                              c.in2_buf == a.in_buf && c.Cin2 == mid && c.stride2 == 1 && c.out_buf != a.in_buf && a.in_buf >= 0 && a.Hi == c.Ho &&
                              a.Wi == c.Wo && b.Hi == a.Hi && b.Ho == a.Hi && c.Hi2 == a.Hi && c.Wi2 == a.Wi && !(flags & SALVE_RESNET_NO_PROJ_FUSE);
            if (!(shapes || proj) || mid != 64 || a.Hi % 8 != 0) continue;
            // the two intermediate tensors are not produced by the fused kernel: nobody may read them afterwards
            bool dead = true;
            for (int which = 0; which < 2 && dead; which++) {
                const int id = which ? b.out_buf : a.out_buf;
                for (size_t k = i + 3; k < h->ops.size(); k++) {
                    const salve_resnet_op_t& o = h->ops[k];
                    if (o.in_buf == id || (o.op == SALVE_OP_CONV && (o.res_buf == id || o.in2_buf == id))) { dead = false; break; }
                    if (o.out_buf == id) break;
                }
            }
            if (dead) { h->fused[i] = shapes ? 1 : 2; i += 2; }
        }
    }
    // (r6) The last fused block of layer 1 takes the following block's first convolution along (bottleneck_kernel's NEXT form): a 1x1 /
    // stride 1 convolution with ReLU, 4 MID -> 2 MID channels, that reads the block's output Y and nothing else.  If Y's only other
    // readers, up to the buffer's next writer, are stride-2 projection shortcuts, only its even pixels are stored.
    h->next.assign(h->ops.size(), 0);
    for (size_t i = 0; !(flags & SALVE_RESNET_NO_NEXT_FUSE) && i + 3 < h->ops.size(); i++) {
        if (h->fused[i] != 1) continue;
        const salve_resnet_op_t &a = h->ops[i], &c = h->ops[i + 2], &n = h->ops[i + 3];
        const int mid = a.Cout;
        const bool fits = n.op == SALVE_OP_CONV && n.KH == 1 && n.KW == 1 && n.stride == 1 && n.pad == 0 && n.relu && n.res_buf == SALVE_NO_BUF &&
                          n.in2_buf == SALVE_NO_BUF && n.in_buf == c.out_buf && n.Cin == 4 * mid && n.Cout == 2 * mid && n.Hi == c.Ho && n.Wi == c.Wo &&
                          n.out_buf >= 0 && n.out_buf != c.out_buf && n.out_buf != a.in_buf && !h->fused[i + 3];
        if (!fits) continue;
        bool even = !(flags & SALVE_RESNET_CHAIN_STORE_ALL) && c.Ho == c.Wo && !(c.Ho & 1), any = false;
        for (size_t k = i + 4; even && k < h->ops.size(); k++) {
            const salve_resnet_op_t& o = h->ops[k];
            if (o.in_buf == c.out_buf || (o.op == SALVE_OP_CONV && o.res_buf == c.out_buf)) even = false;
            if (o.op == SALVE_OP_CONV && o.in2_buf == c.out_buf) {
                if (o.stride2 == 2 && o.Hi2 == c.Ho && o.Wi2 == c.Wo) any = true; else even = false;
            }
            if (o.op != SALVE_OP_AVGPOOL_FC && o.out_buf == c.out_buf) break;
        }
        h->next[i] = (even && any) ? 2 : 1;
    }
    h->chain.assign(h->ops.size(), 0);
    {
        // 2 (default) = expand_chain_kernel wherever the shapes allow, chained with the next block's first convolution where
        // that one follows directly; 1 = the expand convolution only; 0 = the implicit-GEMM kernels
        const int mode = (flags & SALVE_RESNET_NO_CHAIN) ? 0 : ((flags & SALVE_RESNET_CHAIN_EXPAND_ONLY) ? 1 : 2);
        for (size_t i = 0; mode > 0 && i < h->ops.size(); i++) {
            const salve_resnet_op_t& c = h->ops[i];
            if (c.op != SALVE_OP_CONV || h->fused[i] || (i >= 1 && h->fused[i - 1]) || (i >= 2 && h->fused[i - 2]) || h->stem[i]) continue;
            const bool expand = c.KH == 1 && c.KW == 1 && c.stride == 1 && c.pad == 0 && c.relu && c.res_buf != SALVE_NO_BUF &&
                                c.in2_buf == SALVE_NO_BUF && c.Cout == 4 * c.Cin && expand_shape(c.Cin) && c.res_buf != c.out_buf &&
                                c.in_buf != c.out_buf && c.in_buf >= 0 && c.res_buf >= 0;
            if (!expand) continue;
            h->chain[i] = 1;
            if (mode >= 2 && i + 1 < h->ops.size()) {
                const salve_resnet_op_t& a = h->ops[i + 1];
                const bool next = a.op == SALVE_OP_CONV && a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.relu &&
                                  a.res_buf == SALVE_NO_BUF && a.in2_buf == SALVE_NO_BUF && a.in_buf == c.out_buf && a.Cin == c.Cout &&
                                  a.Hi == c.Ho && a.Wi == c.Wo && chain_shape(c.Cin, a.Cout) && a.out_buf != c.out_buf &&
                                  a.out_buf != c.in_buf && a.out_buf != c.res_buf && !h->fused[i + 1];
                if (next) h->chain[i] = 2;
            }
        }
        // (r5) A chained block whose output Y is otherwise read only by a stride-2 projection shortcut (the last block of a stage: the
        // next block's first convolution is computed in the kernel, its shortcut samples Y at even rows and columns) stores only those
        // pixels: 3/4 of Y's bytes never cross HBM.  Every reader of the buffer up to its next writer is checked.
        h->y_even.assign(h->ops.size(), 0);
        for (size_t i = 0; i < h->ops.size(); i++) {
            if (h->chain[i] != 2 || (flags & SALVE_RESNET_CHAIN_STORE_ALL)) continue;
            const salve_resnet_op_t& c = h->ops[i];
            if (c.Ho != c.Wo || (c.Ho & 1) || h->ops[i + 1].Cout == c.Cin) continue;   // expand_chain_kernel: YEVEN = MIDN != MID
            bool ok = true, any = false;
            for (size_t k = i + 2; k < h->ops.size(); k++) {
                const salve_resnet_op_t& o = h->ops[k];
                if (o.in_buf == c.out_buf || (o.op == SALVE_OP_CONV && o.res_buf == c.out_buf)) { ok = false; break; }
                if (o.op == SALVE_OP_CONV && o.in2_buf == c.out_buf) {
                    if (o.stride2 == 2 && o.Hi2 == c.Ho && o.Wi2 == c.Wo) any = true; else { ok = false; break; }
                }
                if (o.op != SALVE_OP_AVGPOOL_FC && o.out_buf == c.out_buf) break;
            }
            if (ok && any) h->y_even[i] = 1;
        }
        h->chain_waves = (flags & SALVE_RESNET_CHAIN_16_WAVES) ? 16 : 8;
        h->chain_split = (flags & SALVE_RESNET_CHAIN_NO_SPLIT) ? 0 : 1;
#ifdef SALVE_BUILD_ABLATIONS   // development build only: timing-only switches of expand_chain_kernel (they compute WRONG results)
        if (const char* d = getenv("SALVE_CHAIN_DBG")) h->chain_dbg = atoi(d);
#endif
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            h->n_cus = cus;
    }
    if (hipMalloc(&h->d_weights, weights_bytes) != hipSuccess || hipMalloc(&h->d_params, params_bytes) != hipSuccess ||
        hipMalloc(&h->d_ktab, ktab_entries * sizeof(int32_t)) != hipSuccess || hipMalloc(&h->d_zeros, 256) != hipSuccess) {
        salve_fail("salve_resnet_create: hipMalloc failed");
        salve_resnet_destroy(h);
        return nullptr;
    }
    if (hipMemcpy(h->d_weights, weights_f16, weights_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_params, params_f32, params_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_ktab, ktab, ktab_entries * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(h->d_zeros, 0, 256) != hipSuccess) {
        salve_fail("salve_resnet_create: hipMemcpy failed");
        salve_resnet_destroy(h);
        return nullptr;
    }
    return h;
}

void salve_resnet_destroy(void* handle) {
    ResnetHandle* h = reinterpret_cast<ResnetHandle*>(handle);
    if (!h) return;
    if (h->d_weights) (void)hipFree(h->d_weights);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_ktab) (void)hipFree(h->d_ktab);
    if (h->d_zeros) (void)hipFree(h->d_zeros);
    delete h;
}

int salve_resnet_num_layers(void* handle) { return handle ? reinterpret_cast<ResnetHandle*>(handle)->num_layers : 0; }

size_t salve_resnet_workspace_bytes(void* handle, int32_t batch) {
    ResnetHandle* h = reinterpret_cast<ResnetHandle*>(handle);
    if (!h || batch <= 0) return 0;
    return (size_t)h->n_bufs * (size_t)batch * h->max_act_elems * sizeof(uint16_t) + 256;
}

int salve_resnet_forward(void* handle, const void* input, int32_t batch, float* logits, void* workspace,
                         size_t workspace_bytes, int32_t* status, void* stream) {
    ResnetHandle* h = reinterpret_cast<ResnetHandle*>(handle);
    if (!h || !input || !logits || !workspace || batch <= 0) {
        salve_fail("salve_resnet_forward: null argument or bad batch");
        return SALVE_ERR_BAD_ARG;
    }
    if (workspace_bytes < salve_resnet_workspace_bytes(handle, batch)) {
        salve_fail("salve_resnet_forward: workspace too small");
        return SALVE_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    uint16_t* base = reinterpret_cast<uint16_t*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const size_t buf_elems = (size_t)batch * h->max_act_elems;
    auto buf = [&](int i) -> uint16_t* { return i < 0 ? const_cast<uint16_t*>(reinterpret_cast<const uint16_t*>(input)) : base + (size_t)i * buf_elems; };
    for (size_t oi = 0; oi < h->ops.size(); oi++) {
        const salve_resnet_op_t& o = h->ops[oi];
        if (o.op == SALVE_OP_CONV && h->stem[oi]) {
            const salve_resnet_op_t& pool = h->ops[oi + 1];
            StemArgs a;
            a.x = buf(o.in_buf);
            a.w = h->d_weights + o.w_off;
            a.bias = h->d_params + o.b_off;
            a.y = buf(pool.out_buf);
            a.zeros = h->d_zeros;
            a.B = batch; a.H = o.Hi; a.W = o.Wi; a.status = status;
            const long long strips = (long long)batch * ((o.Hi / 4) / STEM_R);
            const unsigned grid = (unsigned)(strips < h->n_cus ? strips : h->n_cus);   // persistent: one workgroup per CU, a contiguous range of strips each
            if (o.Cin == 24) hipLaunchKernelGGL(stem_pool_kernel<3>, dim3(grid), dim3(STEM_THREADS), 0, s, a);
            else if (o.Cin == 16) hipLaunchKernelGGL(stem_pool_kernel<2>, dim3(grid), dim3(STEM_THREADS), 0, s, a);
            else hipLaunchKernelGGL(stem_pool_kernel<1>, dim3(grid), dim3(STEM_THREADS), 0, s, a);
            SALVE_HIP_CHECK(hipGetLastError());
            oi += 1;
            continue;
        }
        if (o.op == SALVE_OP_CONV && h->fused[oi]) {
            const salve_resnet_op_t &ob = h->ops[oi + 1], &oc = h->ops[oi + 2];
            BottleneckArgs a;
            a.x = buf(o.in_buf);
            a.y = buf(oc.out_buf);
            a.wa = h->d_weights + o.w_off; a.wb = h->d_weights + ob.w_off; a.wc = h->d_weights + oc.w_off;
            a.ba = h->d_params + o.b_off; a.bb = h->d_params + ob.b_off; a.bc = h->d_params + oc.b_off;
            a.zeros = h->d_zeros;
            a.B = batch; a.H = o.Hi; a.W = o.Wi; a.status = status; a.xcd_contig = h->xcd_contig;
            const bool narrow = o.Cout == 64;  // 64 mid channels: 8 x 16 pixel tiles; 128: 4 x 16
            const int th = narrow ? 8 : 4;
            // a strip of exactly TH columns right of the whole tile columns (56 = 3 x 16 + 8) is covered by transposed tiles of 16 rows
            const bool strip = (o.Wi % 16) == th && !h->no_transposed_tiles;
            a.tiles_x = strip ? o.Wi / 16 : (o.Wi + 15) / 16;
            a.tiles_y = o.Hi / th;
            a.tiles_t = strip ? (o.Hi + 15) / 16 : 0;
            const long long grid = (long long)batch * (a.tiles_x * a.tiles_y + a.tiles_t);
            if (grid > 0x7FFFFFFFll) { salve_fail("batch too large"); return SALVE_ERR_BAD_ARG; }
            a.wd = nullptr; a.bd = nullptr; a.t1n = nullptr; a.y_even = 0;
            const bool with_next = h->fused[oi] == 1 && narrow && h->next[oi];
            if (with_next) {
                const salve_resnet_op_t& on = h->ops[oi + 3];
                a.wd = h->d_weights + on.w_off; a.bd = h->d_params + on.b_off; a.t1n = buf(on.out_buf); a.y_even = h->next[oi] == 2;
            }
            if (h->fused[oi] == 2) hipLaunchKernelGGL((bottleneck_kernel<64, 8, true>), dim3((unsigned)grid), dim3(BN_THREADS), 0, s, a);
            else if (with_next) hipLaunchKernelGGL((bottleneck_kernel<64, 8, false, true>), dim3((unsigned)grid), dim3(BN_THREADS), 0, s, a);
            else if (narrow) hipLaunchKernelGGL((bottleneck_kernel<64, 8>), dim3((unsigned)grid), dim3(BN_THREADS), 0, s, a);
            else hipLaunchKernelGGL((bottleneck_kernel<128, 4>), dim3((unsigned)grid), dim3(BN_THREADS), 0, s, a);
            SALVE_HIP_CHECK(hipGetLastError());
            oi += with_next ? 3 : 2;
            continue;
        }
        if (o.op == SALVE_OP_CONV && h->chain[oi]) {
            const bool chained = h->chain[oi] == 2;
            const salve_resnet_op_t& on = h->ops[chained ? oi + 1 : oi];
            ChainArgs a;
            a.t2 = buf(o.in_buf); a.x = buf(o.res_buf); a.y = buf(o.out_buf);
            a.wc = h->d_weights + o.w_off; a.bc = h->d_params + o.b_off;
            a.t1n = chained ? buf(on.out_buf) : nullptr;
            a.wa = chained ? h->d_weights + on.w_off : nullptr;
            a.ba = chained ? h->d_params + on.b_off : nullptr;
            a.zeros = h->d_zeros;
            const long long M = (long long)batch * o.Ho * o.Wo;
            if (M > 0x7FFFFFFFll) { salve_fail("batch too large"); return SALVE_ERR_BAD_ARG; }
            a.M = (int)M;
            a.status = status;
            a.y_even_w = h->y_even[oi] ? o.Wo : 0;
            a.dbg = h->chain_dbg;
            const int mid = o.Cin, midn = chained ? on.Cout : 0;
            // SALVE_CHAIN_WAVES=16: 16 waves, tiles of 256 pixels, where the registers allow it (<= 128 per lane).  Four waves per SIMD
            // issue more densely than two -- with every memory operation switched off the layer-2 launch takes 1.22 instead of
            // 1.55 ms -- but with them it takes the same 1.95 ms: at 4.1 TB/s of mixed reads and writes in 64-byte row pieces the
            // memory system is what is left.  Bit-identical; not the default.
            const bool wide = h->chain_waves == 16 && mid == 128 && (!chained || midn == 128);
            // 16 waves with the channels split over wave pairs (128-pixel tiles) for the 256-channel shapes: they are bound by
            // their instruction issue at two waves per SIMD (236-246 VGPRs), not by memory
            const bool split = !wide && h->chain_split && (mid == 256 || (chained && midn == 256));
            const int rows = wide ? 256 : 128;
            a.n_tiles = (int)((M + rows - 1) / rows);
            const unsigned grid = (unsigned)(a.n_tiles < h->n_cus ? a.n_tiles : h->n_cus);   // persistent: one workgroup per CU
            const dim3 blk((wide || split) ? 1024 : 512);
            if (split && chained && mid == 128) hipLaunchKernelGGL((expand_chain_kernel<128, 256, 8, true, 2, 16, 2>), dim3(grid), blk, 0, s, a);
            else if (split && chained) hipLaunchKernelGGL((expand_chain_kernel<256, 256, 5, true, 2, 16, 2>), dim3(grid), blk, 0, s, a);
            else if (split) hipLaunchKernelGGL((expand_chain_kernel<256, 256, 12, false, 2, 16, 2>), dim3(grid), blk, 0, s, a);
            else if (wide && chained) hipLaunchKernelGGL((expand_chain_kernel<128, 128, 6, true, 2, 16>), dim3(grid), blk, 0, s, a);
            else if (wide) hipLaunchKernelGGL((expand_chain_kernel<128, 128, 7, false, 2, 16>), dim3(grid), blk, 0, s, a);
            else if (chained && mid == 128 && midn == 128) hipLaunchKernelGGL((expand_chain_kernel<128, 128, 12, true>), dim3(grid), blk, 0, s, a);
            else if (chained && mid == 128 && midn == 256) hipLaunchKernelGGL((expand_chain_kernel<128, 256, 8, true>), dim3(grid), blk, 0, s, a);
            else if (chained && mid == 256 && midn == 256) hipLaunchKernelGGL((expand_chain_kernel<256, 256, 5, true>), dim3(grid), blk, 0, s, a);
            else if (!chained && mid == 128) hipLaunchKernelGGL((expand_chain_kernel<128, 128, 14, false>), dim3(grid), blk, 0, s, a);
            else if (!chained && mid == 256) hipLaunchKernelGGL((expand_chain_kernel<256, 256, 12, false>), dim3(grid), blk, 0, s, a);
            else { salve_fail("expand_chain: unsupported shape"); return SALVE_ERR_UNSUPPORTED; }
            SALVE_HIP_CHECK(hipGetLastError());
            if (chained) oi += 1;
            continue;
        }
        if (o.op == SALVE_OP_CONV) {
            ConvArgs a;
            a.in = buf(o.in_buf);
            a.w = h->d_weights + o.w_off;
            a.bias = h->d_params + o.b_off;
            a.res = o.res_buf != SALVE_NO_BUF ? buf(o.res_buf) : nullptr;
            a.out = buf(o.out_buf);
            a.ktab = h->d_ktab + o.ktab_off;
            a.zeros = h->d_zeros;
            a.status = status;
            a.xcd_contig = h->xcd_contig;
            a.B = batch; a.Hi = o.Hi; a.Wi = o.Wi; a.Cin = o.Cin; a.Ho = o.Ho; a.Wo = o.Wo; a.Cout = o.Cout;
            a.stride = o.stride; a.pad = o.pad; a.K = o.KH * o.KW * o.Cin; a.relu = o.relu;
            const bool src2 = o.in2_buf != SALVE_NO_BUF;
            a.in2 = src2 ? buf(o.in2_buf) : nullptr;
            a.Hi2 = o.Hi2; a.Wi2 = o.Wi2; a.Cin2 = o.Cin2; a.stride2 = o.stride2;
            a.nkt1 = a.K / BK;
            if (src2) a.K += o.Cin2;
            const long long M = (long long)batch * o.Ho * o.Wo;
            if (M > 0x7FFFFFFFll) { salve_fail("batch too large"); return SALVE_ERR_BAD_ARG; }
            a.M = (int)M;
            a.KW = o.KW;
            a.cin_log2 = 0;
            while ((1 << a.cin_log2) < o.Cin) a.cin_log2++;
            // (the 256 x 256 tiles of the 8-phase kernel are a quarter as many workgroups, one per CU: small launches stay on
            //  conv_igemm_kernel -- the results are bit-identical either way)
            // Measured on ResNet-50 (tools/measure/bench_resnet.py): +1 % for the whole forward at batch 4096, -1 % at 2048 and below
            // (one workgroup per CU: a launch of fewer than six rounds loses more in its last round than the kernel gains).
            const bool few_tiles = h->wide[oi] == WIDE_8PHASE && h->wide_auto && ((M + 255) / 256) * (long long)(o.Cout / 256) < 1536;
            if (h->wide[oi] != WIDE_OFF && !few_tiles) {
                const int cfg = h->wide[oi];
                const int bn = (cfg == WIDE_256_K64_S2 || cfg == WIDE_8PHASE) ? 256 : 128;
                a.m_tiles = (int)((M + C8_BM - 1) / C8_BM);
                a.n_tiles = o.Cout / bn;
                const unsigned grid = (unsigned)(((a.m_tiles + 7) / 8) * 8 * a.n_tiles);
                const bool pw = o.KH == 1 && o.KW == 1 && o.stride == 1 && o.pad == 0;
#ifdef SALVE_BUILD_ABLATIONS
#define WIDE_LAUNCH(BN_, KS_, NS_)                                                                                                  \
    {                                                                                                                               \
        if (src2) hipLaunchKernelGGL((conv_wide_kernel<BN_, KS_, NS_, true, true>), dim3(grid), dim3(WIDE_THREADS), 0, s, a);       \
        else if (pw) hipLaunchKernelGGL((conv_wide_kernel<BN_, KS_, NS_, true, false>), dim3(grid), dim3(WIDE_THREADS), 0, s, a);   \
        else hipLaunchKernelGGL((conv_wide_kernel<BN_, KS_, NS_, false, false>), dim3(grid), dim3(WIDE_THREADS), 0, s, a);          \
    }
#endif
                if (cfg == WIDE_8PHASE) {
                    if (src2) hipLaunchKernelGGL((conv8_kernel<true, true>), dim3(grid), dim3(C8_THREADS), 0, s, a);
                    else if (pw) hipLaunchKernelGGL((conv8_kernel<true, false>), dim3(grid), dim3(C8_THREADS), 0, s, a);
                    else hipLaunchKernelGGL((conv8_kernel<false, false>), dim3(grid), dim3(C8_THREADS), 0, s, a);
                }
#ifdef SALVE_BUILD_ABLATIONS
                else if (cfg == WIDE_PC_128) {
                    if (src2) hipLaunchKernelGGL((conv_pc_kernel<true, true>), dim3(grid), dim3(PC_THREADS), 0, s, a);
                    else if (pw) hipLaunchKernelGGL((conv_pc_kernel<true, false>), dim3(grid), dim3(PC_THREADS), 0, s, a);
                    else hipLaunchKernelGGL((conv_pc_kernel<false, false>), dim3(grid), dim3(PC_THREADS), 0, s, a);
                } else if (cfg == WIDE_256_K64_S2) WIDE_LAUNCH(256, 64, 2)
                else WIDE_LAUNCH(128, 64, 1)
#undef WIDE_LAUNCH
#endif
                SALVE_HIP_CHECK(hipGetLastError());
                continue;
            }
            a.m_tiles = (int)((M + BM - 1) / BM);
            const int bn = (o.Cout % 128 == 0) ? 128 : 64;
            a.n_tiles = o.Cout / bn;
            const unsigned grid = (unsigned)(((a.m_tiles + 7) / 8) * 8 * a.n_tiles);
            const bool pointwise = o.KH == 1 && o.KW == 1 && o.stride == 1 && o.pad == 0;
            if (src2 && bn == 128) {
                hipLaunchKernelGGL((conv_igemm_kernel<128, true, true>), dim3(grid), dim3(CONV_THREADS), 0, s, a);
            } else if (src2) {
                hipLaunchKernelGGL((conv_igemm_kernel<64, true, true>), dim3(grid), dim3(CONV_THREADS), 0, s, a);
            } else if (bn == 128 && pointwise) {
                hipLaunchKernelGGL((conv_igemm_kernel<128, true>), dim3(grid), dim3(CONV_THREADS), 0, s, a);
            } else if (bn == 128) {
                hipLaunchKernelGGL((conv_igemm_kernel<128, false>), dim3(grid), dim3(CONV_THREADS), 0, s, a);
            } else if (pointwise) {
                hipLaunchKernelGGL((conv_igemm_kernel<64, true>), dim3(grid), dim3(CONV_THREADS), 0, s, a);
            } else {
                hipLaunchKernelGGL((conv_igemm_kernel<64, false>), dim3(grid), dim3(CONV_THREADS), 0, s, a);
            }
        } else if (o.op == SALVE_OP_MAXPOOL) {
            const long long total = (long long)batch * o.Ho * o.Wo * (o.Cin / 8);
            hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, buf(o.in_buf), buf(o.out_buf),
                               batch, o.Hi, o.Wi, o.Cin, o.Ho, o.Wo);
        } else {
            hipLaunchKernelGGL(avgpool_fc_kernel, dim3(batch), dim3(256), 0, s, buf(o.in_buf), o.Hi * o.Wi,
                               o.Cin, h->d_params + o.w_off, h->d_params + o.b_off, o.Cout, logits);
        }
        SALVE_HIP_CHECK(hipGetLastError());
    }
    return SALVE_OK;
}

}  // extern "C"
