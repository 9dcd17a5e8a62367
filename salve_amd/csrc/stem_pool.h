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
//   * the (2 R + 1) x (W / 2) convolution rows go to LDS as fp16 (over the patch, which is dead by then), are pooled there
//     and only the R x (W / 4) x 64 pooled outputs are written: 0.2 GB instead of 0.8 + 0.8 + 0.2.
// Price: one convolution row in (2 R + 1) is computed twice (it belongs to the strip above as well): 12.5 % for R = 4.
// Rounding: each convolution output is rounded to fp16 before the max, exactly as the two-kernel path stores it -- the
// pooled tensor is bit-identical (tests/test_gpu_verifier.py).
typedef __attribute__((__ext_vector_type__(8))) _Float16 half8;
constexpr int STEM_THREADS = 512;
constexpr int STEM_R = 4;              // pooled rows per strip
constexpr int STEM_MAX_W = 224;        // input width the static LDS array is sized for

struct StemArgs {
    const uint16_t* x;      // [B, H, W, 8] fp16
    const uint16_t* w;      // [64][7][8][8] fp16 (BatchNorm folded)
    const float* bias;      // [64]
    uint16_t* y;            // [B, H / 4, W / 4, 64] fp16
    const uint16_t* zeros;
    int B, H, W;
    int32_t* status;
    int xcd_contig;   // 1: every XCD owns a contiguous range of strips (resnet.hip: xcd_linear): neighbouring strips share 7 of their 23 input rows
};

__global__ __launch_bounds__(STEM_THREADS, 2) void stem_pool_kernel(StemArgs p) {
    constexpr int R = STEM_R, NCR = 2 * R + 1, NPR = 4 * R + 7;
    constexpr int PATCH_E = NPR * (STEM_MAX_W + 6) * 8;        // uint16 elements
    constexpr int W_E = 7 * 64 * 64;
    __shared__ __attribute__((aligned(1024))) uint16_t smem[PATCH_E + W_E];
    uint16_t* patch = smem;
    uint16_t* wl = smem + PATCH_E;
    uint16_t* stage = smem;                                    // [NCR][Wo][64] fp16, written after the MFMAs

    const int W = p.W, H = p.H, PC = W + 6, Wo = W >> 1, Ho = H >> 1, Wp = Wo >> 1, Hp = Ho >> 1;
    const int strips = Hp / R;
    const int blk = xcd_linear(blockIdx.x, gridDim.x, p.xcd_contig);
    const int b = blk / strips, s = blk % strips;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frag_row = lane & 15, frag_q = lane >> 4;
    const int iy_base = 4 * R * s - 5;                         // input row of patch row 0
    const int cr_base = 2 * R * s - 1;                         // convolution row of local row 0

    // ---- fill: weights (7 k-tiles of 64 rows x 128 bytes, source-side swizzle as in conv_igemm_kernel), then the patch
    {
        const int row_base = tid >> 3;
        const int chunk = (tid & 7) ^ ((row_base >> 1) & 7);
        const uint16_t* wsrc = p.w + (long long)row_base * 448 + chunk * 8;
#pragma unroll
        for (int kt = 0; kt < 7; kt++)
            __builtin_amdgcn_global_load_lds((global_cptr)(wsrc + kt * 64), (lds_ptr)(wl + kt * 4096 + wave * 512), 16, 0, 0);
        const int total = NPR * PC;                            // 16-byte units of the patch, row-major
        const uint16_t* ximg = p.x + (long long)b * H * W * 8;
#if defined(STEM_NO_LOAD)
        for (int i0 = 0; i0 < STEM_THREADS; i0 += STEM_THREADS) {
#else
        for (int i0 = 0; i0 < total; i0 += STEM_THREADS) {
#endif
            const int i = i0 + tid;
            if (i < total) {
                const int prow = i / PC, pcol = i - prow * PC;
                const int iy = iy_base + prow, ix = pcol - 3;
                const uint16_t* src = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? ximg + ((long long)iy * W + ix) * 8 : p.zeros;
                __builtin_amdgcn_global_load_lds((global_cptr)src, (lds_ptr)(patch + (i0 + wave * 64) * 8), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- MFMAs: wave w owns the 16-pixel tiles w, w + 8, ... of the NCR x (Wo / 16) tiles of the strip
    const int tiles_x = Wo >> 4, n_tiles = NCR * tiles_x;      // 7, 63
    f32x4 acc[8][4];
    int abase[8];                                              // patch offset (elements) of the lane's pixel, k-chunk q
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int t = wave + 8 * i;
        if (t >= n_tiles) t = n_tiles - 1;                     // a duplicate, discarded below
        const int crl = t / tiles_x, cx = (t - crl * tiles_x) * 16 + frag_row;
        abase[i] = ((2 * crl) * PC + 2 * cx + frag_q) * 8;
    }
    // the accumulators start at zero and the bias is added at the end, as in conv_igemm_kernel (same rounding)
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bsw = (frag_row >> 1) & 7;
#if defined(STEM_NO_MFMA)
    for (int kh = 0; kh < 0; kh++) {
#else
    for (int kh = 0; kh < 7; kh++) {
#endif
#pragma unroll
        for (int h = 0; h < 2; h++) {
            act8 af[8], bfr[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                bfr[j] = *reinterpret_cast<const act8*>(wl + kh * 4096 + (j * 16 + frag_row) * 64 + (((h * 4 + frag_q) ^ bsw) * 8));
#pragma unroll
            for (int i = 0; i < 8; i++) af[i] = *reinterpret_cast<const act8*>(patch + abase[i] + (kh * PC + 4 * h) * 8);
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();   // everyone is done reading the patch and the weights: the staging area overlays them

    // ---- bias + ReLU, one rounding to fp16, into the staging rows
    float amax = 0.f;
#if defined(STEM_NO_EPI)
    if (acc[0][0][0] == 12345.678f)
#endif
    {
        float4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bias[j] = *reinterpret_cast<const float4*>(p.bias + j * 16 + 4 * frag_q);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int t = wave + 8 * i;
            if (t >= n_tiles) continue;
            const int crl = t / tiles_x, cx = (t - crl * tiles_x) * 16 + frag_row;
            uint16_t* px_row = stage + ((crl * Wo + cx) << 6);
            const int rot = cx & 7;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = j * 16 + 4 * frag_q;
                const uint2 o = pack4<true>(amax, acc[i][j] + vec4(bias[j]));
                // the 16-byte chunks of a pixel's 128-byte row are rotated by the pixel's column: the 16 lanes of a store
                // group (16 consecutive pixels, same channels) would otherwise all hit one bank
                *reinterpret_cast<uint2*>(px_row + ((((n >> 3) ^ rot) << 3) | (n & 7))) = o;
            }
        }
    }
    report_range(p.status, amax);
    __syncthreads();

    // ---- 3x3 / stride 2 / pad 1 max-pool out of the staging rows: 8 channels (16 bytes) per thread and pooled pixel
    uint16_t* yimg = p.y + (long long)b * Hp * Wp * 64;
    // one pooled row per pass: thread -> (pooled column, group of 8 channels); W / 4 * 8 <= 512 threads
#if defined(STEM_NO_POOL)
    for (int pr = 0; pr < 0; pr++) {
#else
    for (int pr = 0; pr < R; pr++) {
#endif
        const int cg = tid & 7, px = tid >> 3;
        if (px >= Wp) continue;
        const int py = R * s + pr;
        // the staged values are ReLU outputs (>= 0, no NaN): the packed fp16 maximum is exact, four instructions per window
        // element instead of eight conversions and eight fp32 maxima
        half8 best = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int crl = 2 * pr + dy, cr = cr_base + crl;
            if (cr < 0 || cr >= Ho) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const int cx = 2 * px - 1 + dx;
                if (cx < 0 || cx >= Wo) continue;
                const half8 v = *reinterpret_cast<const half8*>(stage + ((crl * Wo + cx) << 6) + ((cg ^ (cx & 7)) << 3));
                best = __builtin_elementwise_max(best, v);
            }
        }
        const uint4 o4 = __builtin_bit_cast(uint4, best);
        const uint32_t o[4] = {o4.x, o4.y, o4.z, o4.w};
        *reinterpret_cast<uint4*>(yimg + ((long long)py * Wp + px) * 64 + cg * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}
