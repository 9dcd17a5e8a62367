// conv_wide.h -- ALTERNATIVE implicit-GEMM convolution kernels of the verifier (included by resnet.hip inside its
// namespace).  Selected with SALVE_CONV_WIDE=d|e|f; the default network does not use them (resnet.hip: choose_wide).
//
// Same GEMM as conv_igemm_kernel (out[m, n] = sum_k A[m, k] W[n, k], m = (b, oy, ox), k = (kh, kw, ci), NHWC fp16, fp32
// accumulation, v_mfma_f32_16x16x32_f16 with swapped operands, same k order: bit-identical results) on a 256 x BN block
// tile with 8 waves around a RING of LDS slots that global_load_lds fills ahead of the MFMAs:
//   d  256 x 256, KS 64, two slots   (one workgroup per CU; the next stage is in flight under the MFMAs of this one:
//      counted s_waitcnt vmcnt + raw s_barrier, all LDS in one array, no ordinary global load in the loop -- the im2col
//      tap of a stage is wave-uniform arithmetic, not a table read -- stages past the end of K are dummy loads of the zero
//      page so that the counted wait stays a constant)
//   e  256 x 128, KS 64, one slot    (two workgroups per CU hide each other's fills, conv_igemm_kernel's structure)
//   f  256 x 128, KS 64, three slots, 16 waves with SPLIT ROLES (conv_pc_kernel below)
// What they were built to test, and what was measured (MI355X, ResNet-50 shapes at batch 512): halving
// the fill bytes per FLOP (d), deeper prefetch (KS 32 x 4 slots, removed: 64-byte row segments), and separating the
// LDS-DMA issue from the MFMA issue (f) each leave the convolutions at 650-850 TFLOP/s, within +-10 % of conv_igemm_kernel.
// The ablation builds (flags below) say why: with a convolution's MFMAs alone the kernel takes 48-57 % of its time, with
// the fills alone 65 %, with the fragment reads + MFMAs 67-75 %: the three streams ADD instead of overlapping, and the fill
// stream itself cannot exceed about 50 GB/s per CU (13 TB/s) for this gather out of L2 / Infinity Cache.
//
// LDS image of a slot: A rows then W rows, KS halves per row, unpadded (an LDS-DMA wave-instruction writes 1 KiB
// lane-linearly); bank conflicts of the ds_read_b128 fragment reads are removed by a swizzle applied on the SOURCE side:
// position q of row r holds k-chunk q ^ f(r), f(r) = (r >> 1) & 7 for 128-byte rows (KS = 64) and (-(r >> 2)) & 3 for
// 64-byte rows (KS = 32); both are conflict-free for gfx950's 16-lane read groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...

// Development ablations (tools/build_ablations.sh): WIDE_NO_MFMA keeps the fragment reads live but issues no MFMA,
// WIDE_NO_LOADS fills the ring once (stage 0 data reused), WIDE_NO_DSREAD multiplies constant fragments.
#if defined(WIDE_NO_MFMA)
#define WIDE_MFMA(B_, A_, C_) { asm volatile("" ::"v"(B_), "v"(A_)); }
#else
#define WIDE_MFMA(B_, A_, C_) { C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(B_, A_, C_, 0, 0, 0); }
#endif
#if defined(WIDE_NO_DSREAD)
#define WIDE_FRAG(PTR_) (act8{(_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4, (_Float16)5, (_Float16)6, (_Float16)7, (_Float16)8})
#else
#define WIDE_FRAG(PTR_) (*reinterpret_cast<const act8*>(PTR_))
#endif

constexpr int WIDE_THREADS = 512;
constexpr int WIDE_BM = 256;

template <int KS>
__device__ __forceinline__ int wide_swz(int row) {
    return KS == 64 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3);
}

template <int BN, int KS, int NSLOT, bool POINTWISE, bool SRC2>
__global__ __launch_bounds__(WIDE_THREADS, (NSLOT * (WIDE_BM + BN) * KS * 2 <= 80 * 1024 && (WIDE_BM * (BN + 8) * 2) <= 80 * 1024) ? 4 : 2)
void conv_wide_kernel(ConvArgs p) {
    constexpr int BM = WIDE_BM;
    constexpr int CPR = KS / 8;                 // 16-byte chunks per LDS row
    constexpr int RPP = WIDE_THREADS / CPR;     // rows filled per pass of the workgroup
    constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP;
    constexpr int LPS = A_LOADS + B_LOADS;      // LDS-DMA instructions per thread and stage
    constexpr int PF = NSLOT > 1 ? NSLOT - 1 : 1;  // stages in flight (NSLOT 1: unused)
    constexpr int SLOT_E = (BM + BN) * KS;      // uint16 elements per slot
    constexpr int WAVES_N = BN == 256 ? 4 : 2, WAVES_M = 8 / WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile: 128 x 64 (BN 256), 64 x 64 (BN 128)
    constexpr int MI = WM / 16, NJ = WN / 16;
    constexpr int LDC = BN + 8;
    constexpr int RING_E = NSLOT * SLOT_E, C_E = BM * LDC;
    static_assert(B_LOADS >= 1 && BN % RPP == 0, "tile / thread mapping");
    __shared__ __attribute__((aligned(1024))) uint16_t smem[RING_E > C_E ? RING_E : C_E];

    int m_tile, n_tile;
    if (!xcd_tile(blockIdx.x, p.m_tiles, p.n_tiles, p.xcd_contig, m_tile, n_tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int row_base = tid / CPR, cpos = tid % CPR;
    const int chunk = cpos ^ wide_swz<KS>(row_base);   // k-chunk of the stage this thread fetches (RPP is a multiple of 16)

    // per-thread rows of the A tile (fixed for the whole K loop)
    int iy0[A_LOADS], ix0[A_LOADS];
    long long boff[A_LOADS];
    const uint16_t* rowp[A_LOADS];
    const uint16_t* rowp2[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; i++) {
        const int m = m0 + row_base + i * RPP;
        const bool valid = m < p.M;
        rowp2[i] = nullptr;
        rowp[i] = nullptr;
        iy0[i] = ix0[i] = 0;
        boff[i] = 0;
        if (SRC2 && valid) {
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            rowp2[i] = p.in2 + (((long long)b * p.Hi2 + (long long)oy * p.stride2) * p.Wi2 + (long long)ox * p.stride2) * p.Cin2 + chunk * 8;
        }
        if (POINTWISE) {
            rowp[i] = valid ? p.in + (long long)m * p.Cin + chunk * 8 : nullptr;
        } else {
            const int mm = valid ? m : 0;
            const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            iy0[i] = valid ? oy * p.stride - p.pad : -100000;  // rows beyond M read zeros
            ix0[i] = ox * p.stride - p.pad;
            boff[i] = (long long)b * p.Hi * p.Wi;
        }
    }
    const uint16_t* wrow = p.w + (long long)(n0 + row_base) * p.K + chunk * 8;
    const int nst = p.K / KS;          // stages of this convolution
    const int nst1 = p.nkt1 * (64 / KS);  // SRC2: stages from nst1 on read the second source (nkt1 counts 64-deep tiles)

#if defined(WIDE_NO_LOADS)
#define WIDE_SKIP_LOADS(S) if ((S) >= NSLOT) break;
#else
#define WIDE_SKIP_LOADS(S)
#endif
    // Stage S -> slot SLOT (both wave-uniform).  Stages >= nst are dummy loads of the zero page.
#define WIDE_ISSUE(S, SLOT)                                                                                            \
    do {                                                                                                               \
        uint16_t* As_ = smem + (SLOT) * SLOT_E + wave * 64 * 8;                                                        \
        uint16_t* Bs_ = As_ + BM * KS;                                                                                 \
        const bool live_ = (S) < nst;                                                                                  \
        WIDE_SKIP_LOADS(S)                                                                                             \
        const int k0_ = (S) * KS;                                                                                      \
        int dy_ = 0, dx_ = 0, c0_ = 0;                                                                                 \
        if (!POINTWISE) {                                                                                              \
            const int tap_ = k0_ >> p.cin_log2;                                                                        \
            c0_ = k0_ & (p.Cin - 1);                                                                                   \
            dy_ = tap_ / p.KW;                                                                                         \
            dx_ = tap_ - dy_ * p.KW;                                                                                   \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < A_LOADS; i++) {                                                          \
            const uint16_t* src = p.zeros;                                                                             \
            if (live_) {                                                                                               \
                if (SRC2 && (S) >= nst1) {                                                                             \
                    if (rowp2[i]) src = rowp2[i] + ((S) - nst1) * KS;                                                  \
                } else if (POINTWISE) {                                                                                \
                    if (rowp[i]) src = rowp[i] + k0_;                                                                  \
                } else {                                                                                               \
                    const int iy = iy0[i] + dy_, ix = ix0[i] + dx_;                                                    \
                    if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)                                                  \
                        src = p.in + ((boff[i] + (long long)iy * p.Wi + ix) * p.Cin + c0_ + chunk * 8);                \
                }                                                                                                      \
            }                                                                                                          \
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(As_ + i * RPP * KS), 16, 0, 0);               \
        }                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; j++) {                                                          \
            const uint16_t* src = live_ ? wrow + (long long)j * RPP * p.K + k0_ : p.zeros;                             \
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(Bs_ + j * RPP * KS), 16, 0, 0);               \
        }                                                                                                              \
    } while (0)

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frag_row = lane & 15, frag_q = lane >> 4;
    const int frag_sw = wide_swz<KS>(frag_row);

    if constexpr (NSLOT == 1) {
        // one slot: fill, wait, multiply -- the latency is hidden by the other workgroups of the CU (conv_igemm_kernel's
        // structure on the larger tile)
        for (int t = 0; t < nst; t++) {
            WIDE_ISSUE(t, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const uint16_t* As = smem;
            const uint16_t* Bs = As + BM * KS;
#pragma unroll
            for (int ks = 0; ks < KS / 32; ks++) {
                act8 af[MI], bfr[NJ];
                const int pos = ((ks * 4 + frag_q) ^ frag_sw) * 8;
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    bfr[j] = WIDE_FRAG(Bs + (wn * WN + j * 16 + frag_row) * KS + pos);
#pragma unroll
                for (int i = 0; i < MI; i++)
                    af[i] = WIDE_FRAG(As + (wm * WM + i * 16 + frag_row) * KS + pos);
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++)
                        WIDE_MFMA(bfr[j], af[i], acc[i][j])
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
    // prologue: PF stages in flight, then wait for the first one
#pragma unroll
    for (int s = 0; s < PF; s++) { WIDE_ISSUE(s, s); }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory");
    __builtin_amdgcn_s_barrier();

    int slot = 0, fill = PF;   // slot of stage t, slot of stage t + PF ( = slot of stage t - 1)
    for (int t = 0; t < nst; t++) {
        WIDE_ISSUE(t + PF, fill);
        const uint16_t* As = smem + slot * SLOT_E;
        const uint16_t* Bs = As + BM * KS;
#pragma unroll
        for (int ks = 0; ks < KS / 32; ks++) {
            act8 af[MI], bfr[NJ];
            const int pos = ((ks * 4 + frag_q) ^ frag_sw) * 8;
#pragma unroll
            for (int j = 0; j < NJ; j++)
                bfr[j] = WIDE_FRAG(Bs + (wn * WN + j * 16 + frag_row) * KS + pos);
#pragma unroll
            for (int i = 0; i < MI; i++)
                af[i] = WIDE_FRAG(As + (wm * WM + i * 16 + frag_row) * KS + pos);
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    WIDE_MFMA(bfr[j], af[i], acc[i][j])
        }
        // the next stage has landed (this wave's part), this wave is done reading the current slot
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        fill = fill + 1 == NSLOT ? 0 : fill + 1;
    }
    }
#undef WIDE_ISSUE
    // the dummy stages still in flight target slots that the epilogue staging overlays
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue (as conv_igemm_kernel): residual tile -> LDS, bias / residual / ReLU in fp32 on the accumulator's own
    //      elements, one rounding to fp16, 16-byte coalesced stores
    uint16_t* Cs = smem;
    constexpr int CH_PER_ROW = BN / 8;
    constexpr int C_ITERS = (BM * CH_PER_ROW) / WIDE_THREADS;
    float4 bias_v[NJ];   // all bias loads in flight together (one round trip, not one per tile column)
#pragma unroll
    for (int j = 0; j < NJ; j++) bias_v[j] = *reinterpret_cast<const float4*>(p.bias + n0 + wn * WN + j * 16 + 4 * frag_q);
    if (p.res) {
        // batches of 8 loads in flight, then their 8 LDS stores (a load-store pair per iteration would be 16 dependent
        // round trips to L2 / HBM per thread)
        constexpr int RB = C_ITERS < 8 ? C_ITERS : 8;
#pragma unroll
        for (int it0 = 0; it0 < C_ITERS; it0 += RB) {
            uint4 rv[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int id = tid + (it0 + u) * WIDE_THREADS;
                const int m = m0 + id / CH_PER_ROW;
                const long long off = (long long)(m < p.M ? m : 0) * p.Cout + n0 + (id % CH_PER_ROW) * 8;
                rv[u] = *reinterpret_cast<const uint4*>(p.res + off);
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int id = tid + (it0 + u) * WIDE_THREADS;
                *reinterpret_cast<uint4*>(Cs + (id / CH_PER_ROW) * LDC + (id % CH_PER_ROW) * 8) = rv[u];
            }
        }
        __syncthreads();
    }
    float amax = 0.f, amin = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int ncol = wn * WN + j * 16 + 4 * frag_q;  // this lane's 4 consecutive channels of tile column j
        const float4 bias = bias_v[j];
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const int mrow = wm * WM + i * 16 + frag_row;
            uint2* cell = reinterpret_cast<uint2*>(Cs + mrow * LDC + ncol);
            f32x4 v = acc[i][j] + vec4(bias);
            if (p.res) v += vec4(*cell);
            *cell = pack4_lo(amax, amin, v, p.relu ? 0.f : -65504.f);
        }
    }
    report_range(p.status, p.relu ? amax : fmaxf(amax, -amin));
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < C_ITERS; it++) {
        const int id = tid + it * WIDE_THREADS;
        const int r = id / CH_PER_ROW, ch = id % CH_PER_ROW;
        const int m = m0 + r;
        if (m < p.M)
            *reinterpret_cast<uint4*>(p.out + (long long)m * p.Cout + n0 + ch * 8) = *reinterpret_cast<const uint4*>(Cs + r * LDC + ch * 8);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_pc_kernel -- the same ring with the work SPLIT BY WAVE ROLE: 16 waves, waves 0-7 multiply (ds_read + MFMA only, no
// vector-memory instruction in their loop), waves 8-15 only fill the ring (address arithmetic + global_load_lds).
// Why: the ablations of conv_wide_kernel show its three streams -- LDS-DMA issue with its address
// arithmetic, fragment reads, MFMAs -- adding up instead of overlapping: a wave that issues a stage's 6-8 LDS-DMA
// instructions (150-200 cycles each with the im2col arithmetic) cannot issue MFMAs meanwhile, and the two waves of a SIMD
// do it at the same time after every barrier.  With the roles split, the matrix pipe of a SIMD is fed by two consumer waves
// while two loader waves on the same SIMD issue the fills.
// Tile 256 x 128 (consumers 4 x 2, 64 x 64 each: the allocation of all 16 waves must stay within 128 registers), KS = 64,
// three slots of 48 KB, one workgroup per CU.
constexpr int PC_THREADS = 1024;

template <bool POINTWISE, bool SRC2>
__global__ __launch_bounds__(PC_THREADS, 4) void conv_pc_kernel(ConvArgs p) {
    constexpr int BM = 256, BN = 128, KS = 64, NSLOT = 3, PF = NSLOT - 1;
    constexpr int RPP = 64;                       // rows filled per pass of the 512 loader threads (8 chunks per row)
    constexpr int A_LOADS = BM / RPP, B_LOADS = BN / RPP, LPS = A_LOADS + B_LOADS;
    constexpr int SLOT_E = (BM + BN) * KS;
    constexpr int WM = 64, WN = 64, MI = 4, NJ = 4;
    constexpr int LDC = BN + 8;
    constexpr int RING_E = NSLOT * SLOT_E, C_E = BM * LDC;
    __shared__ __attribute__((aligned(1024))) uint16_t smem[RING_E > C_E ? RING_E : C_E];

    int m_tile, n_tile;
    if (!xcd_tile(blockIdx.x, p.m_tiles, p.n_tiles, p.xcd_contig, m_tile, n_tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int nst = p.K / KS;
    const int frag_row = lane & 15, frag_q = lane >> 4;
    f32x4 acc[MI][NJ];
    const int cw = wave & 7, wm = cw >> 1, wn = cw & 1;   // consumer wave position (loaders: unused)

    if (wave >= 8) {
        // ------------------------------------------------------------------ loaders
        const int lt = tid - 512, lwave = wave - 8;
        const int row_base = lt >> 3, cpos = lt & 7;
        const int chunk = cpos ^ wide_swz<KS>(row_base);
        int iy0[A_LOADS], ix0[A_LOADS];
        long long boff[A_LOADS];
        const uint16_t* rowp[A_LOADS];
        const uint16_t* rowp2[A_LOADS];
#pragma unroll
        for (int i = 0; i < A_LOADS; i++) {
            const int m = m0 + row_base + i * RPP;
            const bool valid = m < p.M;
            rowp2[i] = nullptr;
            rowp[i] = nullptr;
            iy0[i] = ix0[i] = 0;
            boff[i] = 0;
            if (SRC2 && valid) {
                const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
                rowp2[i] = p.in2 + (((long long)b * p.Hi2 + (long long)oy * p.stride2) * p.Wi2 + (long long)ox * p.stride2) * p.Cin2 + chunk * 8;
            }
            if (POINTWISE) {
                rowp[i] = valid ? p.in + (long long)m * p.Cin + chunk * 8 : nullptr;
            } else {
                const int mm = valid ? m : 0;
                const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, b = t / p.Ho;
                iy0[i] = valid ? oy * p.stride - p.pad : -100000;
                ix0[i] = ox * p.stride - p.pad;
                boff[i] = (long long)b * p.Hi * p.Wi;
            }
        }
        const uint16_t* wrow = p.w + (long long)(n0 + row_base) * p.K + chunk * 8;
        const int nst1 = p.nkt1;
#define PC_ISSUE(S, SLOT)                                                                                              \
    do {                                                                                                               \
        uint16_t* As_ = smem + (SLOT) * SLOT_E + lwave * 64 * 8;                                                       \
        uint16_t* Bs_ = As_ + BM * KS;                                                                                 \
        const bool live_ = (S) < nst;                                                                                  \
        const int k0_ = (S) * KS;                                                                                      \
        int dy_ = 0, dx_ = 0, c0_ = 0;                                                                                 \
        if (!POINTWISE) {                                                                                              \
            const int tap_ = k0_ >> p.cin_log2;                                                                        \
            c0_ = k0_ & (p.Cin - 1);                                                                                   \
            dy_ = tap_ / p.KW;                                                                                         \
            dx_ = tap_ - dy_ * p.KW;                                                                                   \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < A_LOADS; i++) {                                                          \
            const uint16_t* src = p.zeros;                                                                             \
            if (live_) {                                                                                               \
                if (SRC2 && (S) >= nst1) {                                                                             \
                    if (rowp2[i]) src = rowp2[i] + ((S) - nst1) * KS;                                                  \
                } else if (POINTWISE) {                                                                                \
                    if (rowp[i]) src = rowp[i] + k0_;                                                                  \
                } else {                                                                                               \
                    const int iy = iy0[i] + dy_, ix = ix0[i] + dx_;                                                    \
                    if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)                                                  \
                        src = p.in + ((boff[i] + (long long)iy * p.Wi + ix) * p.Cin + c0_ + chunk * 8);                \
                }                                                                                                      \
            }                                                                                                          \
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(As_ + i * RPP * KS), 16, 0, 0);               \
        }                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < B_LOADS; j++) {                                                          \
            const uint16_t* src = live_ ? wrow + (long long)j * RPP * p.K + k0_ : p.zeros;                             \
            __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(Bs_ + j * RPP * KS), 16, 0, 0);               \
        }                                                                                                              \
    } while (0)
#pragma unroll
        for (int s = 0; s < PF; s++) { PC_ISSUE(s, s); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory");
        __builtin_amdgcn_s_barrier();
        int fill = PF;
        for (int t = 0; t < nst; t++) {
            PC_ISSUE(t + PF, fill);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PF - 1) * LPS) : "memory");   // stage t + 1 has landed
            __builtin_amdgcn_s_barrier();
            fill = fill + 1 == NSLOT ? 0 : fill + 1;
        }
#undef PC_ISSUE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy stages target slots the epilogue staging overlays
    } else {
        // ------------------------------------------------------------------ consumers
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int frag_sw = wide_swz<KS>(frag_row);
        __builtin_amdgcn_s_barrier();   // stage 0 is in its slot
        int slot = 0;
        for (int t = 0; t < nst; t++) {
            const uint16_t* As = smem + slot * SLOT_E;
            const uint16_t* Bs = As + BM * KS;
#pragma unroll
            for (int ks = 0; ks < KS / 32; ks++) {
                act8 af[MI], bfr[NJ];
                const int pos = ((ks * 4 + frag_q) ^ frag_sw) * 8;
#pragma unroll
                for (int j = 0; j < NJ; j++) bfr[j] = *reinterpret_cast<const act8*>(Bs + (wn * WN + j * 16 + frag_row) * KS + pos);
#pragma unroll
                for (int i = 0; i < MI; i++) af[i] = *reinterpret_cast<const act8*>(As + (wm * WM + i * 16 + frag_row) * KS + pos);
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // done reading this slot
            __builtin_amdgcn_s_barrier();
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
    }
    __builtin_amdgcn_s_barrier();   // ring quiet: loaders have drained their queue, consumers have read everything

    // ---- epilogue: all 16 waves move data, the 8 consumer waves own the accumulators
    uint16_t* Cs = smem;
    constexpr int CH_PER_ROW = BN / 8;
    constexpr int C_ITERS = (BM * CH_PER_ROW) / PC_THREADS;
    if (p.res) {
        uint4 rv[C_ITERS];
#pragma unroll
        for (int u = 0; u < C_ITERS; u++) {
            const int id = tid + u * PC_THREADS;
            const int m = m0 + id / CH_PER_ROW;
            rv[u] = *reinterpret_cast<const uint4*>(p.res + (long long)(m < p.M ? m : 0) * p.Cout + n0 + (id % CH_PER_ROW) * 8);
        }
#pragma unroll
        for (int u = 0; u < C_ITERS; u++) {
            const int id = tid + u * PC_THREADS;
            *reinterpret_cast<uint4*>(Cs + (id / CH_PER_ROW) * LDC + (id % CH_PER_ROW) * 8) = rv[u];
        }
        __syncthreads();
    }
    if (wave < 8) {
        float amax = 0.f, amin = 0.f;
        float4 bias_v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) bias_v[j] = *reinterpret_cast<const float4*>(p.bias + n0 + wn * WN + j * 16 + 4 * frag_q);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int ncol = wn * WN + j * 16 + 4 * frag_q;
            const float4 bias = bias_v[j];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                const int mrow = wm * WM + i * 16 + frag_row;
                uint2* cell = reinterpret_cast<uint2*>(Cs + mrow * LDC + ncol);
                f32x4 v = acc[i][j] + vec4(bias);
                if (p.res) v += vec4(*cell);
                *cell = pack4_lo(amax, amin, v, p.relu ? 0.f : -65504.f);
            }
        }
        report_range(p.status, p.relu ? amax : fmaxf(amax, -amin));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < C_ITERS; u++) {
        const int id = tid + u * PC_THREADS;
        const int r = id / CH_PER_ROW, ch = id % CH_PER_ROW;
        const int m = m0 + r;
        if (m < p.M)
            *reinterpret_cast<uint4*>(p.out + (long long)m * p.Cout + n0 + ch * 8) = *reinterpret_cast<const uint4*>(Cs + r * LDC + ch * 8);
    }
}
