// conv8.h -- implicit-GEMM convolution on a 256 x 256 tile with an EIGHT-PHASE software pipeline (included by resnet.hip
// inside its namespace).  Same GEMM, operands, k order and epilogue as conv_igemm_kernel / conv_wide_kernel: bit-identical
// results (tests/test_gpu_verifier.py).
//
// Why: the 128 x 128 kernel with one LDS stage and four workgroups per CU sits at the ceiling of its structure (DESIGN.md
// section 4.4: the LDS-DMA fill of its tiles alone caps the matrix pipes at 38 %); a 256 x 256 tile halves the fill bytes per
// FLOP but leaves one workgroup per CU, whose fills, fragment reads and MFMAs then ADD unless the workgroup overlaps them
// itself (conv_wide.h, configuration d).  This kernel does, after the 256^2 template of cdna_hip_programming.md section 5:
//
//   * 8 waves, wave (wr, wc) = (wave >> 2, wave & 3) owns 128 pixels x 64 channels = 8 x 4 accumulator tiles (128 VGPRs).
//   * A k-tile (64 deep) is FOUR half-tiles of 128 rows x 64 halves (16 KB): A-h0 / A-h1 hold the first / second 64 pixels
//     of every wave row, B-h0 / B-h1 the first / second 32 channels of every wave column.  LDS holds two k-tiles (128 KB).
//   * A k-tile is multiplied in four PHASES, one accumulator quadrant each: (A0,B0) (A0,B1) (A1,B1) (A1,B0); the fragments
//     of a half-tile are read from LDS once (A0 and B0 before phase 0, B1 before phase 1, A1 before phase 2).
//   * Every phase stages exactly ONE half-tile (two global_load_lds per thread), five or six phases before its first read:
//     phase (kt, 0) B-h1(kt+1), (kt, 1) A-h1(kt+1), (kt, 2) A-h0(kt+2), (kt, 3) B-h0(kt+2); its slot's previous content was
//     last read two or three phases earlier.  The wait is COUNTED: s_waitcnt vmcnt(8) after the phase's own two loads
//     retires the half-tile staged four phases ago -- which is first read in the NEXT phase, behind two barriers.
//   * A phase is  [fragment reads, staging, vmcnt(8)]  s_barrier  [lgkmcnt(0), 16 MFMAs]  s_barrier.  Waves 4-7 (the SIMD
//     partners of waves 0-3) run ONE BARRIER BEHIND: while one wave of a SIMD multiplies, the other reads fragments and
//     issues its LDS-DMA.  The margins above hold with that stagger (RAW: the lagging group's wait for a half-tile ends at
//     the barrier in front of the leading group's first read; WAR: the lagging group's last read of a slot is complete
//     one barrier before the leading group restages it).
//   * Stagings past the end of K are dummy loads of the zero page, so the counted wait stays a constant.
// The im2col tap of a k-tile is wave-uniform arithmetic (Cin is a multiple of 64 here), rows keep a pointer to their tap
// (0, 0) pixel and add one offset per k-tile.

constexpr int C8_THREADS = 512;
constexpr int C8_BM = 256, C8_BN = 256, C8_KS = 64;
constexpr int C8_HALF_E = 128 * C8_KS;   // uint16 elements of a half-tile

// (timing-only builds -- no MFMAs / no fills / fills never waited for: tools/probe/ablations/timing_switches.patch)
#define C8_MFMA(B_, A_, C_) { C_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(B_, A_, C_, 0, 0, 0); }

template <bool POINTWISE, bool SRC2>
__global__ __launch_bounds__(C8_THREADS, 2) void conv8_kernel(ConvArgs p) {
    constexpr int BM = C8_BM, BN = C8_BN, KS = C8_KS;
    constexpr int LDC = BN + 8;
    constexpr int RING_E = 8 * C8_HALF_E, C_E = BM * LDC;
    __shared__ __attribute__((aligned(1024))) uint16_t smem[RING_E > C_E ? RING_E : C_E];

    int m_tile, n_tile;   // XCD-aware (resnet.hip: xcd_tile)
    if (!xcd_tile(blockIdx.x, p.m_tiles, p.n_tiles, p.xcd_contig, m_tile, n_tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int r8 = tid >> 3;                                  // 0..63: row of a half-tile this thread stages (and + 64)
    const int chunk = (tid & 7) ^ ((r8 >> 1) & 7);            // its k-chunk (source-side swizzle; the same for r8 + 64)

    // the four A rows (pixels) of this thread: [s][i] = row (r8 + 64 i) of half-tile A-h<s> = pixel i * 128 + s * 64 + r8
    int iy0[2][2], ix0[2][2];
    const uint16_t* rowp[2][2];
    const uint16_t* rowp2[2][2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int m = m0 + i * 128 + s * 64 + r8;
            const bool valid = m < p.M;
            rowp[s][i] = nullptr;
            rowp2[s][i] = nullptr;
            iy0[s][i] = ix0[s][i] = 0;
            if (SRC2 && valid) {
                const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
                rowp2[s][i] = p.in2 + (((long long)b * p.Hi2 + (long long)oy * p.stride2) * p.Wi2 + (long long)ox * p.stride2) * p.Cin2 + chunk * 8;
            }
            if (POINTWISE) {
                rowp[s][i] = valid ? p.in + (long long)m * p.Cin + chunk * 8 : nullptr;
            } else {
                const int mm = valid ? m : 0;
                const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, b = t / p.Ho;
                iy0[s][i] = valid ? oy * p.stride - p.pad : -100000;  // rows beyond M read zeros
                ix0[s][i] = ox * p.stride - p.pad;
                rowp[s][i] = p.in + (((long long)b * p.Hi + (valid ? iy0[s][i] : 0)) * p.Wi + ix0[s][i]) * p.Cin + chunk * 8;
            }
        }
    // the four weight rows: row (r8 + 64 i) of half-tile B-h<s> = channel ((r8 + 64 i) >> 5) * 64 + s * 32 + (r8 & 31)
    const uint16_t* wrow[2][2];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int i = 0; i < 2; i++)
            wrow[s][i] = p.w + (long long)(n0 + ((r8 + 64 * i) >> 5) * 64 + s * 32 + (r8 & 31)) * p.K + chunk * 8;

    const int nkt = p.K / KS;

    // half-tile kinds in a buffer: 0 = A-h0, 1 = B-h0, 2 = B-h1, 3 = A-h1
#define C8_SLOT(KT, KIND) (smem + ((((KT) & 1) << 2) + (KIND)) * C8_HALF_E)
    // stage half-tile KIND of k-tile KT (wave-uniform KT; past the end of K: the zero page)
#define C8_STAGE_GUARD(KT)
#define C8_STAGE(KT, KIND)                                                                                             \
    do {                                                                                                               \
        C8_STAGE_GUARD(KT)                                                                                             \
        uint16_t* dst_ = C8_SLOT(KT, KIND) + wave * 8 * KS;                                                            \
        const int kt_ = (KT);                                                                                          \
        const bool live_ = kt_ < nkt;                                                                                  \
        constexpr int s_ = ((KIND) == 2 || (KIND) == 3) ? 1 : 0;                                                       \
        if ((KIND) == 0 || (KIND) == 3) {                                                                              \
            int dy_ = 0, dx_ = 0, delta_ = 0;                                                                          \
            if (!POINTWISE) {                                                                                          \
                const int k0_ = kt_ * KS;                                                                              \
                const int tap_ = k0_ >> p.cin_log2;                                                                    \
                dy_ = tap_ / p.KW;                                                                                     \
                dx_ = tap_ - dy_ * p.KW;                                                                               \
                delta_ = (dy_ * p.Wi + dx_) * p.Cin + (k0_ & (p.Cin - 1));                                             \
            }                                                                                                          \
            _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                            \
                const uint16_t* src = p.zeros;                                                                         \
                if (live_) {                                                                                           \
                    if (SRC2 && kt_ >= p.nkt1) {                                                                       \
                        if (rowp2[s_][i]) src = rowp2[s_][i] + (kt_ - p.nkt1) * KS;                                    \
                    } else if (POINTWISE) {                                                                            \
                        if (rowp[s_][i]) src = rowp[s_][i] + kt_ * KS;                                                 \
                    } else if ((unsigned)(iy0[s_][i] + dy_) < (unsigned)p.Hi && (unsigned)(ix0[s_][i] + dx_) < (unsigned)p.Wi) { \
                        src = rowp[s_][i] + delta_;                                                                    \
                    }                                                                                                  \
                }                                                                                                      \
                __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(dst_ + i * 64 * KS), 16, 0, 0);           \
            }                                                                                                          \
        } else {                                                                                                       \
            _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                            \
                const uint16_t* src = live_ ? wrow[s_][i] + kt_ * KS : p.zeros;                                        \
                __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(dst_ + i * 64 * KS), 16, 0, 0);           \
            }                                                                                                          \
        }                                                                                                              \
    } while (0)

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frag_row = lane & 15, frag_q = lane >> 4;
    const int frag_sw = (frag_row >> 1) & 7;   // ((row >> 1) & 7) of a fragment row: the wave / tile offsets are multiples of 16
    // fragment read offsets inside a half-tile (elements): row * KS + swizzled chunk
    const int a_off = (wr * 64 + frag_row) * KS, b_off = (wc * 32 + frag_row) * KS;
    const int pos0 = ((0 * 4 + frag_q) ^ frag_sw) * 8, pos1 = ((1 * 4 + frag_q) ^ frag_sw) * 8;

    act8 af[4][2], bf0[2][2], bf1[2][2];
#define C8_READ_A(KT, KIND)                                                                                            \
    {                                                                                                                  \
        const uint16_t* h_ = C8_SLOT(KT, KIND) + a_off;                                                                \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                \
            af[i][0] = *reinterpret_cast<const act8*>(h_ + i * 16 * KS + pos0);                                        \
            af[i][1] = *reinterpret_cast<const act8*>(h_ + i * 16 * KS + pos1);                                        \
        }                                                                                                              \
    }
#define C8_READ_B(DST, KT, KIND)                                                                                       \
    {                                                                                                                  \
        const uint16_t* h_ = C8_SLOT(KT, KIND) + b_off;                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                                \
            DST[j][0] = *reinterpret_cast<const act8*>(h_ + j * 16 * KS + pos0);                                       \
            DST[j][1] = *reinterpret_cast<const act8*>(h_ + j * 16 * KS + pos1);                                       \
        }                                                                                                              \
    }
#define C8_PRIO(V) __builtin_amdgcn_s_setprio(V)
#define C8_MULT(QA, BF, QB)                                                                                            \
    {                                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        C8_PRIO(1);                                                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                               \
            _Pragma("unroll") for (int i = 0; i < 4; i++)                                                              \
                _Pragma("unroll") for (int j = 0; j < 2; j++) C8_MFMA(BF[j][ks], af[i][ks], acc[(QA) * 4 + i][(QB) * 2 + j]) \
        C8_PRIO(0);                                                                                                    \
    }
#define C8_VMWAIT() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define C8_WAIT_BARRIER()                                                                                              \
    {                                                                                                                  \
        C8_VMWAIT();                                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                  \
    }
    // Variants measured and dropped (l3 / l4 shapes of ResNet-50 at batch 4096, 2.96 ms for four of them with this schedule):
    // the phase's two LDS-DMA instructions in the shadow of its MFMAs, after the first eight, with the counted wait in front
    // of them (+4 %); the staging in front of the fragment reads (+2 %); no s_setprio around the MFMAs (+11 %); staging
    // spread by the fragment reads' free time -- none in phase 0, two half-tiles in phase 3 (+1 %).
#define C8_PHASE_HEAD(READS, STAGE) { READS; STAGE; }

    // prologue: what phases -6 .. -1 would have staged; the first two half-tiles have landed before anybody reads
    C8_STAGE(0, 0);
    C8_STAGE(0, 1);
    C8_STAGE(0, 2);
    C8_STAGE(0, 3);
    C8_STAGE(1, 0);
    C8_STAGE(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();   // the stagger: waves 4-7 run one barrier behind from here on
    for (int kt = 0; kt < nkt; kt++) {
        // ---- phase 0: quadrant (A0, B0)
        C8_PHASE_HEAD({ C8_READ_A(kt, 0); C8_READ_B(bf0, kt, 1); }, C8_STAGE(kt + 1, 2));
        C8_WAIT_BARRIER();
        C8_MULT(0, bf0, 0);
        __builtin_amdgcn_s_barrier();
        // ---- phase 1: quadrant (A0, B1)
        C8_PHASE_HEAD({ C8_READ_B(bf1, kt, 2); }, C8_STAGE(kt + 1, 3));
        C8_WAIT_BARRIER();
        C8_MULT(0, bf1, 1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2: quadrant (A1, B1)
        C8_PHASE_HEAD({ C8_READ_A(kt, 3); }, C8_STAGE(kt + 2, 0));
        C8_WAIT_BARRIER();
        C8_MULT(1, bf1, 1);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3: quadrant (A1, B0)
        C8_STAGE(kt + 2, 1);
        C8_WAIT_BARRIER();
        C8_MULT(1, bf0, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();    // waves 0-3 wait for the others' last phase
#undef C8_STAGE
#undef C8_READ_A
#undef C8_READ_B
#undef C8_MULT
#undef C8_WAIT_BARRIER
#undef C8_PHASE_HEAD
#undef C8_SLOT
    // the dummy stagings still in flight target slots that the epilogue staging overlays
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue (as conv_igemm_kernel): residual tile -> LDS, bias / residual / ReLU in fp32 on the accumulator's own
    //      elements, one rounding to fp16, 16-byte coalesced stores
    uint16_t* Cs = smem;
    constexpr int CH_PER_ROW = BN / 8;
    constexpr int C_ITERS = (BM * CH_PER_ROW) / C8_THREADS;
    float4 bias_v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bias_v[j] = *reinterpret_cast<const float4*>(p.bias + n0 + wc * 64 + j * 16 + 4 * frag_q);
    // (one 64-bit row offset per thread, then a scalar step per iteration: a v_mad_i64_i32 per copy otherwise)
    constexpr int ROW_STEP = C8_THREADS / CH_PER_ROW;
    const int crow = tid / CH_PER_ROW, cch = tid % CH_PER_ROW;
    const long long coff = (long long)(m0 + crow) * p.Cout + n0 + cch * 8;
    const int cstep = ROW_STEP * p.Cout;
    if (p.res) {
        constexpr int RB = 8;
#pragma unroll
        for (int it0 = 0; it0 < C_ITERS; it0 += RB) {
            uint4 rv[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const bool in = m0 + crow + (it0 + u) * ROW_STEP < p.M;   // (rows beyond M read row m0: any valid address, the value is not stored)
                rv[u] = *reinterpret_cast<const uint4*>(p.res + (in ? coff + (long long)(it0 + u) * cstep : (long long)m0 * p.Cout + n0));
            }
#pragma unroll
            for (int u = 0; u < RB; u++)
                *reinterpret_cast<uint4*>(Cs + (crow + (it0 + u) * ROW_STEP) * LDC + cch * 8) = rv[u];
        }
        __syncthreads();
    }
    float amax = 0.f;
    float amin = 0.f;
    const float lo = p.relu ? 0.f : -65504.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ncol = wc * 64 + j * 16 + 4 * frag_q;  // this lane's 4 consecutive channels of tile column j
        const float4 bias = bias_v[j];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int mrow = wr * 128 + i * 16 + frag_row;
            uint2* cell = reinterpret_cast<uint2*>(Cs + mrow * LDC + ncol);
            f32x4 v = acc[i][j] + vec4(bias);
            if (p.res) v += vec4(*cell);
            *cell = pack4_lo(amax, amin, v, lo);
        }
    }
    if (!p.relu) amax = fmaxf(amax, -amin);
    report_range(p.status, amax);
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < C_ITERS; it++) {
        if (m0 + crow + it * ROW_STEP < p.M)
            *reinterpret_cast<uint4*>(p.out + coff + (long long)it * cstep) = *reinterpret_cast<const uint4*>(Cs + (crow + it * ROW_STEP) * LDC + cch * 8);
    }
}

// A variant that was built and dropped (conv8b): the WEIGHT fragments read straight from global memory into registers -- a
// lane's fragment is 16 contiguous bytes of one weight row -- which takes the B half-tiles out of the LDS traffic (160 KB
// instead of 256 KB per k-tile) and would, by the ablations above, be worth up to a third.  It cannot be scheduled with counted
// waits: LDS-DMA operations and ordinary vector loads share the vmcnt counter but do NOT complete in order with respect to
// each other, so "all but the last N" does not say WHICH operations have landed (with hand-counted waits the kernel computed
// garbage, mostly in the lagging wave group; hipcc knows -- it puts s_waitcnt vmcnt(0) in front of the first use of an
// ordinarily loaded register while LDS-DMA is outstanding, and with those drains the kernel is correct but no faster than
// the one-stage kernels: a drain waits out the latency of whatever was issued last).  Staging A through registers as well
// (all loads of one kind) does not fit the 256 VGPRs next to 128 accumulators.
