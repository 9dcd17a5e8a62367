// stem_pool.h -- the verifier's stem as ONE kernel: 7x7 / stride 2 / pad 3 convolution (+ folded BatchNorm + ReLU) and the
// 3x3 / stride 2 / pad 1 max-pool behind it (early_fusion.py:67-71), for the 8-channel (two images) input.  Included by
// resnet.hip inside its namespace.
//
// Why a kernel of its own: as an implicit GEMM the stem re-fetches every input pixel 12 times (7 x 7 taps / stride^2) --
// 5.75 GB of L2 -> LDS fill at batch 512, at the ~13 TB/s this gather reaches that alone is 440 us --, writes a 112 x 112 x 64
// tensor (0.8 GB) that the pooling kernel reads back, and leaves the MFMAs waiting.  Here a workgroup owns a STRIP of R pooled
// rows over the full image width:
//   * the input patch the strip needs ((4 R + 7) rows x (W + 6) columns x 16 bytes, zero padded) is brought into LDS once
//     (84 KB for R = 4 at 224 x 224: every input pixel is fetched 1.4 times instead of 12), all 57 KB of weights next to it;
//   * the A fragments of the MFMAs are read STRAIGHT out of the patch: an output pixel's k-chunk (kh, kw) is input pixel
//     (2 oy + kh, 2 ox + kw), 16 bytes; 16 consecutive output pixels are 16 lanes 32 bytes apart -- conflict-free for
//     gfx950's 16-lane read groups; no im2col copy, no barrier inside the K loop, so the two waves of a SIMD drift apart and
//     one's fragment reads run under the other's MFMAs;
//   * (round 4) the (2 R + 1) x (W / 2) convolution rows are pooled IN REGISTERS: wave w < 7 owns the 16-pixel column w of
//     eight of the nine rows, wave 7 the middle row of all columns (handed over through 14 KB of LDS), so the vertical maximum
//     is lane-local, the horizontal one two DPP row shifts (the one neighbour beyond a column's left edge comes through LDS),
//     and only the R x (W / 4) x 64 pooled outputs are written: 0.2 GB instead of 0.8 + 0.8 + 0.2.  Rounds 2-3 staged the
//     rows as fp16 over the dead patch and the weights (129 KB written, 258 KB read, per strip) and pooled from there;
//   * (round 4) with nothing overlaying them the weights stay resident: the workgroup is PERSISTENT (one per CU, a contiguous
//     range of strips each), loads them once, and the next strip's patch streams in (LDS-DMA) while this strip's rows are
//     rounded, pooled and stored -- with 142 KB of LDS only one workgroup fits a CU, so nothing else overlaps these phases.
// Price: one convolution row in (2 R + 1) is computed twice (it belongs to the strip above as well): 12.5 % for R = 4.
// Rounding: each convolution output is rounded to fp16 before the max, exactly as the two-kernel path stores it -- the
// pooled tensor is bit-identical (tests/test_gpu_verifier.py).
typedef __attribute__((__ext_vector_type__(8))) _Float16 half8;
typedef __attribute__((__ext_vector_type__(2))) _Float16 half2v;
constexpr int STEM_THREADS = 512;
constexpr int STEM_R = 4;              // pooled rows per strip
constexpr int STEM_W = 224;            // the input width the kernel is written for (compile-time: the patch offsets of the A fragments
                                       // are immediates; other widths take the two-kernel path)

struct StemArgs {
    const uint16_t* x;      // [B, H, W, 8 G] fp16   (G = channel groups of 8: 1 for two images, 2 for four)
    const uint16_t* w;      // [64][G][7][8][8] fp16 (BatchNorm folded): K ordered (group, kh, kw, channel in group)
    const float* bias;      // [64]
    uint16_t* y;            // [B, H / 4, W / 4, 64] fp16
    const uint16_t* zeros;
    int B, H, W;
    int32_t* status;
};

// packed fp16 maximum of ReLU outputs (>= 0, no NaN): exact
__device__ __forceinline__ uint2 stem_max(uint2 a, uint2 b) {
    const half2v x = __builtin_elementwise_max(__builtin_bit_cast(half2v, a.x), __builtin_bit_cast(half2v, b.x));
    const half2v y = __builtin_elementwise_max(__builtin_bit_cast(half2v, a.y), __builtin_bit_cast(half2v, b.y));
    return uint2{__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, y)};
}
// the value of the lane one to the left (CTRL 0x111: row_shr 1) / right (0x101: row_shl 1) inside its 16-lane row, 0 beyond the row
template <int CTRL>
__device__ __forceinline__ uint2 stem_neighbour(uint2 v) {
    return uint2{(uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, 0xF, 0xF, true),
                 (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, 0xF, 0xF, true)};
}

// LDS accesses of the hand-over while the next patch's LDS-DMA is in flight, as inline asm: hipcc puts an `s_waitcnt vmcnt(0)` in
// front of every LDS access it takes for a possible alias of an LDS-DMA destination (resnet.hip: lds_read8) -- here that would wait
// for the whole next patch.  The compiler does not count these: the reads carry their own wait, the writes are drained by the
// explicit `s_waitcnt lgkmcnt(0)` in front of the barriers below.
__device__ __forceinline__ uint32_t stem_lds_addr(const void* p) { return (uint32_t)(unsigned long long)((lds_ptr)p); }
__device__ __forceinline__ void stem_lds_write8(uint32_t addr, uint2 v) {
    const unsigned long long u = ((unsigned long long)v.y << 32) | v.x;
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(u) : "memory");
}
__device__ __forceinline__ uint2 stem_u2(unsigned long long u) { return uint2{(uint32_t)u, (uint32_t)(u >> 32)}; }
// four 8-byte reads at four addresses / at one address + 0, 32, 64, 96 bytes
__device__ __forceinline__ void stem_lds_read8x4(const uint32_t (&addr)[4], uint2 (&out)[4]) {
    unsigned long long r0, r1, r2, r3;
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]) : "memory");
    out[0] = stem_u2(r0); out[1] = stem_u2(r1); out[2] = stem_u2(r2); out[3] = stem_u2(r3);
}
__device__ __forceinline__ void stem_lds_read8x4_strided(uint32_t addr, uint2 (&out)[4]) {
    unsigned long long r0, r1, r2, r3;
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:32\n\tds_read_b64 %2, %4 offset:64\n\tds_read_b64 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
    out[0] = stem_u2(r0); out[1] = stem_u2(r1); out[2] = stem_u2(r2); out[3] = stem_u2(r3);
}
#define STEM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   // (raw: __syncthreads() would drain the patch in flight)

// G = 2 / 3 (r6; the 12-channel early fusion of two surfaces, input padded to 16 channels: BASELINE config 5; 18 -> 24 with the layout modality): the SAME kernel run over the channel
// groups of 8 one after the other -- patch and weights of a group do not fit LDS beside those of the other (2 x (84 + 57) KB), so a strip
// loads group 0's patch + weights, multiplies, loads group 1's into the same LDS, multiplies into the same accumulators, and the next strip's
// group 0 streams in under the epilogue as before.  The host packs the stem's K in that order (group, kh, kw, channel in group) and the
// two-kernel path walks it in the same order through its k table, so the pooled tensor is still bit-identical.  Before: the generic
// convolution + max-pool, 7.4 + 1.7 ms per 4096 samples.  (The G = 1 instantiation is kept TEXTUALLY what it was: the kernel sits at 254
// registers without scratch, and a scratch reload is a vmcnt load queued behind the next patch.)
template <int G>
__global__ __launch_bounds__(STEM_THREADS, 2) void stem_pool_kernel(StemArgs p) {
    constexpr int R = STEM_R, NPR = 4 * R + 7;                 // (2 R + 1 = 9 convolution rows per strip)
    constexpr int PATCH_E = NPR * (STEM_W + 6) * 8;        // uint16 elements
    constexpr int W_E = 7 * 64 * 64;
    constexpr int XT = STEM_W / 32;                        // 16-pixel columns of a convolution row: 7
    constexpr int XROW_E = XT * 16 * 64;                       // the middle row, [column][pixel][64], chunks rotated by the pixel
    constexpr int XEDGE_E = XT * R * 64;                       // the vertical maxima of every column's last pixel, [column][pooled row][64]
    __shared__ __attribute__((aligned(1024))) uint16_t smem[PATCH_E + W_E + XROW_E + XEDGE_E];
    __shared__ __attribute__((aligned(16))) float bias_s[64];   // (in LDS: a global load in the epilogue would queue behind the next patch)
    uint16_t* patch = smem;
    uint16_t* wl = smem + PATCH_E;
    uint16_t* xrow = wl + W_E;
    uint16_t* xedge = xrow + XROW_E;

    constexpr int W = STEM_W, PC = W + 6, Wo = W >> 1, Wp = Wo >> 1, tiles_x = Wo >> 4;
    const int H = p.H, Hp = H >> 2;
    const int strips = Hp / R;
    const long long total_strips = (long long)p.B * strips;
    const long long st_lo = total_strips * blockIdx.x / gridDim.x, st_hi = total_strips * (blockIdx.x + 1) / gridDim.x;
    if (st_lo >= st_hi) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frag_row_ = lane & 15, frag_q_ = lane >> 4;

    // ---- the weights, once (7 k-tiles of 64 rows x 128 bytes, source-side swizzle as in conv_igemm_kernel)
    if constexpr (G == 1) {
        const int row_base = tid >> 3;
        const int chunk = (tid & 7) ^ ((row_base >> 1) & 7);
        const uint16_t* wsrc = p.w + (long long)row_base * 448 + chunk * 8;
#pragma unroll
        for (int kt = 0; kt < 7; kt++)
            __builtin_amdgcn_global_load_lds((global_cptr)(wsrc + kt * 64), (lds_ptr)(wl + kt * 4096 + wave * 512), 16, 0, 0);
    }
    // (G == 2) ... of channel group g, per strip and group
    auto issue_weights = [&](int g) {
        int tl = tid;
        asm volatile("" : "+v"(tl));                           // (opaque: recomputed where it is used, not carried through the MFMA loops)
        const int row_base = tl >> 3;
        const int chunk = (tl & 7) ^ ((row_base >> 1) & 7);
        const uint16_t* wsrc = p.w + (long long)row_base * (448 * G) + g * 448 + chunk * 8;
#pragma unroll
        for (int kt = 0; kt < 7; kt++)
            __builtin_amdgcn_global_load_lds((global_cptr)(wsrc + kt * 64), (lds_ptr)(wl + kt * 4096 + wave * 512), 16, 0, 0);
    };
    if constexpr (G > 1) issue_weights(0);
    // ---- a strip's input patch: (4 R + 7) rows x (W + 6) columns x 16 bytes, zero padded, row-major
    auto issue_patch = [&](int b, int s) {
        const int iy_base = 4 * R * s - 5;                     // input row of patch row 0
        int pc = PC;
        asm volatile("" : "+s"(pc));                           // (opaque: the per-thread patch coordinates of the eleven passes are recomputed
                                                               //  per strip instead of living in registers through the MFMA loop)
        const int total = NPR * pc;
        const uint16_t* ximg = p.x + (long long)b * H * W * 8;
        for (int i0 = 0; i0 < total; i0 += STEM_THREADS) {
            const int i = i0 + tid;
            if (i < total) {
                const int prow = i / pc, pcol = i - prow * pc;
                const int iy = iy_base + prow, ix = pcol - 3;
                const uint16_t* src = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? ximg + ((long long)iy * W + ix) * 8 : p.zeros;
                __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(patch + (i0 + wave * 64) * 8), 16, 0, 0);
            }
        }
    };
    auto issue_patch_g = [&](int b, int s, int g) {           // (G == 2) the same for channel group g of a 16-channel input
        const int iy_base = 4 * R * s - 5;
        int pc = PC;
        asm volatile("" : "+s"(pc));
        const int total = NPR * pc;
        const uint16_t* ximg = p.x + (long long)b * H * W * (8 * G) + 8 * g;
        for (int i0 = 0; i0 < total; i0 += STEM_THREADS) {
            const int i = i0 + tid;
            if (i < total) {
                const int prow = i / pc, pcol = i - prow * pc;
                const int iy = iy_base + prow, ix = pcol - 3;
                const uint16_t* src = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? ximg + ((long long)iy * W + ix) * (8 * G) : p.zeros;
                __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(patch + (i0 + wave * 64) * 8), 16, 0, 0);
            }
        }
    };
    // (image, strip) of the workgroup's current strip: 32-bit counters -- a 64-bit division per strip lives in vector registers
    int b = __builtin_amdgcn_readfirstlane((int)(st_lo / strips)), s = __builtin_amdgcn_readfirstlane((int)(st_lo % strips));
    const int n_mine = __builtin_amdgcn_readfirstlane((int)(st_hi - st_lo));
    if constexpr (G == 1) issue_patch(b, s); else issue_patch_g(b, s, 0);

    // ---- who computes what: wave w < tiles_x the rows 0 .. R-1, R+1 .. 2R of column w, wave 7 row R of every column
    const bool row_wave = wave == 7, col_wave = wave < tiles_x && wave < 7;
    int abase[8];                                              // patch offset (elements) of the lane's pixel, k-chunk q
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int crl = row_wave ? R : (i < R ? i : i + 1);
        int xt = row_wave ? i : wave;
        if (xt >= tiles_x) xt = tiles_x - 1;                   // a duplicate, discarded below
        abase[i] = ((2 * crl) * PC + 2 * (xt * 16 + frag_row_) + frag_q_) * 8;
    }
    if (tid < 16) *reinterpret_cast<float4*>(bias_s + 4 * tid) = *reinterpret_cast<const float4*>(p.bias + 4 * tid);   // visible behind the first strip's barrier
    const int bsw = (frag_row_ >> 1) & 7;
    float amax = 0.f;

    for (int k = 0; k < n_mine; k++) {
        const int s_next = s + 1 < strips ? s + 1 : 0, b_next = s + 1 < strips ? b : b + 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // the patch (first strip: and the weights) has landed

        // ---- MFMAs: no barrier inside the K loop; the accumulators start at zero and the bias is added at the end, as in conv_igemm_kernel
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (G == 1) {
            for (int kh = 0; kh < 7; kh++) {
    #pragma unroll
                for (int h = 0; h < 2; h++) {
                    act8 af[8], bfr[4];
    #pragma unroll
                    for (int j = 0; j < 4; j++)
                        bfr[j] = *reinterpret_cast<const act8*>(wl + kh * 4096 + (j * 16 + frag_row_) * 64 + (((h * 4 + frag_q_) ^ bsw) * 8));
    #pragma unroll
                    for (int i = 0; i < 8; i++) af[i] = *reinterpret_cast<const act8*>(patch + abase[i] + (kh * PC + 4 * h) * 8);
    #pragma unroll
                    for (int i = 0; i < 8; i++)
    #pragma unroll
                        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
            }
        } else {
            // two channel groups, ONE copy of the loop: the second group's patch and weights go into the LDS the first was read from and are
            // multiplied into the same accumulators
#pragma unroll 1
            for (int g = 0; g < G; g++) {
                if (g > 0) {
                    __syncthreads();
                    issue_patch_g(b, s, g);
                    issue_weights(g);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                for (int kh = 0; kh < 7; kh++) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        act8 af[8], bfr[4];
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            bfr[j] = *reinterpret_cast<const act8*>(wl + kh * 4096 + (j * 16 + frag_row_) * 64 + (((h * 4 + frag_q_) ^ bsw) * 8));
#pragma unroll
                        for (int i = 0; i < 8; i++) af[i] = *reinterpret_cast<const act8*>(patch + abase[i] + (kh * PC + 4 * h) * 8);
#pragma unroll
                        for (int i = 0; i < 8; i++)
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();                                       // everyone is done reading the patch
        float4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bias[j] = lds_read_f4(bias_s + j * 16 + 4 * frag_q_);
        if constexpr (G == 1) {
            if (k + 1 < n_mine) issue_patch(b_next, s_next);   // ... so the next one streams in under the rest of this strip
        } else {
            if (k + 1 < n_mine) { issue_patch_g(b_next, s_next, 0); issue_weights(0); }
        }
        {
            // (the LDS addresses of the hand-over are recomputed per strip from an opaque copy of the lane's coordinates: hoisted out
            //  of the strip loop they cost 77 spilled registers, reloaded inside the MFMA loop)
            int frag_row = frag_row_, frag_q = frag_q_;
            asm volatile("" : "+v"(frag_row), "+v"(frag_q));
            const int rot = frag_row & 7;   // xrow: the 16-byte chunks of a pixel's 128 bytes are rotated by the pixel, or the 16 lanes of a group hit one bank
            // ---- bias + ReLU, one rounding to fp16; a column wave folds its eight rows straight into the vertical maxima of the
            //      3x3 / stride 2 / pad 1 max-pool (pooled row pr: convolution rows 2 pr .. 2 pr + 2; the middle row R joins
            //      behind the barrier), the row wave hands its row over
            uint2 v[R][4];
#pragma unroll
            for (int pr = 0; pr < R; pr++)
#pragma unroll
                for (int j = 0; j < 4; j++) v[pr][j] = uint2{0u, 0u};   // (defined on every path: no values carried around the strip loop)
            uint32_t mid_addr[4];   // the lane's pixel of the middle row, channels j * 16 + 4 q .. + 3
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = j * 16 + 4 * frag_q;
                mid_addr[j] = stem_lds_addr(xrow) + 2 * (((((n >> 3) ^ rot) << 3) | (n & 7)) + (frag_row << 6));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint2 t[8];
#pragma unroll
                for (int i = 0; i < 8; i++) t[i] = pack4<true>(amax, acc[i][j] + vec4(bias[j]));
                if (row_wave) {
#pragma unroll
                    for (int i = 0; i < XT; i++) stem_lds_write8(mid_addr[j] + i * (16 * 64 * 2), t[i]);
                } else {
                    // local row 0 of an image's first strip is convolution row -1: not a row (the ReLU outputs are >= 0: zero is neutral)
                    if (s == 0) t[0] = uint2{0u, 0u};
#pragma unroll
                    for (int pr = 0; pr < R; pr++) {
                        uint2 m = uint2{0u, 0u};
#pragma unroll
                        for (int dy = 0; dy < 3; dy++) {
                            const int crl = 2 * pr + dy;
                            if (crl != R) m = stem_max(m, t[crl < R ? crl : crl - 1]);
                        }
                        v[pr][j] = m;
                    }
                }
            }
            STEM_BARRIER();
            const uint32_t edge_addr = stem_lds_addr(xedge) + 2 * (wave * R * 64 + 4 * frag_q);   // + pr * 128 + j * 32
            if (col_wave) {
                uint2 mid[4];
#pragma unroll
                for (int j = 0; j < 4; j++) mid_addr[j] += wave * (16 * 64 * 2);
                stem_lds_read8x4(mid_addr, mid);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int pr = 0; pr < R; pr++) {
                        if (2 * pr <= R && R <= 2 * pr + 2) v[pr][j] = stem_max(v[pr][j], mid[j]);
                        if (frag_row == 15) stem_lds_write8(edge_addr + pr * 128 + j * 32, v[pr][j]);
                    }
            }
            STEM_BARRIER();
            // ... then the horizontal ones (pooled pixel pl of the column: its pixels 2 pl - 1 .. 2 pl + 1) with two row shifts
            if (col_wave) {
                uint16_t* yimg = p.y + (long long)b * Hp * Wp * 64;
#pragma unroll
                for (int pr = 0; pr < R; pr++) {
                    uint2 edge[4] = {uint2{0u, 0u}, uint2{0u, 0u}, uint2{0u, 0u}, uint2{0u, 0u}};
                    if (frag_row == 0 && wave > 0) stem_lds_read8x4_strided(edge_addr - R * 128 + pr * 128, edge);   // (the column to the left)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int n = j * 16 + 4 * frag_q;
                        uint2 left = stem_neighbour<0x111>(v[pr][j]);
                        const uint2 right = stem_neighbour<0x101>(v[pr][j]);
                        if (frag_row == 0) left = edge[j];
                        const uint2 best = stem_max(stem_max(left, v[pr][j]), right);
                        if (!(frag_row & 1))
                            *reinterpret_cast<uint2*>(yimg + ((long long)(R * s + pr) * Wp + wave * 8 + (frag_row >> 1)) * 64 + n) = best;
                    }
                }
            }
        }
        b = b_next; s = s_next;
    }
    report_range(p.status, amax);
}
