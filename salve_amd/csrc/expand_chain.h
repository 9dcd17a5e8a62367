// expand_chain.h -- the HBM-bound seam between two bottleneck blocks as ONE persistent streaming kernel (included by
// resnet.hip inside its namespace):
//
//     Y   = relu(Wc . t2 + bc + X)        the block's last 1x1 convolution (MID -> 4 MID channels) + residual
//     t1' = relu(Wa' . Y + ba')           the NEXT block's first 1x1 convolution (4 MID -> MIDN channels)   [CHAIN]
//
// (torchvision ResNet v1.5 bottleneck, salve/models/resnet_factory.py:26-44; early_fusion.py:72-79 runs the blocks in order.)
// Round 3, measured with PMC counters at batch 4096 (profiles/r03_resnet_traffic*.md): as separate implicit-GEMM launches
// these two ops run at 3.6 - 4.3 TB/s and 1.45 - 1.75 x their HBM time -- the expand convolution waits 62 - 67 % of its
// wave-cycles on a handful of dependent round trips per 128 x 128 tile -- and the second one reads Y back (a quarter of the
// bytes of a block).  Here Y never comes back: a workgroup owns 128 pixels, produces Y in chunks of 32 channels, stores each
// chunk AND multiplies it -- still in LDS -- into t1'.
//
// Structure (one 512-thread workgroup per CU, persistent over its tiles; everything that comes from memory is prefetched
// through LDS rings by LDS-DMA and nothing is ever drained):
//   * The HBM stream of a workgroup is a sequence of 8 KB ITEMS of 128 rows x 32 halves: per tile first the MID / 32 k-steps
//     of its t2 rows (A items), then the 4 MID / 32 chunks of its residual rows (RES items).  A ring of R slots holds the
//     next R - 1 items in flight ((R - 1) x 8 KB per CU: bandwidth x latency).
//   * Weights stream from L2: Wc chunk [32 ch][MID] two RES steps ahead (3 buffers), Wa' chunk [MIDN][32] two ahead (4).
//   * A step consumes one item and has ONE barrier, at its start (the data of the step is visible; the previous step is
//     complete).  Then the LOADER waves (0-3) issue the LDS-DMA operations for later steps -- into the ring slot, the Wc and the
//     Wa' buffer the step BEFORE the previous one was the last to read -- and the STORER waves (4-7) write the previous
//     step's Y chunk to memory; then all 8 waves (wave w owns pixels 16 w ..) compute: GEMM 1' of the PREVIOUS chunk (its
//     slot and its Wa' buffer are still intact: t1' += Ychunk . Wa'chunk^T, K = 32), then for an A item the wave's A fragments
//     of that k-step go to registers (they stay there for the whole tile: GEMM 3 never reads t2 from LDS again), for a RES
//     item GEMM 3 of 16 pixels x 32 channels (K = MID) and bias + residual + ReLU IN PLACE in the ring slot (the slot now
//     holds the Y chunk).  At the end of the step the loaders wait -- counted, never 0 -- for the next step's data.
//     (First version, two barriers per step with the issue after the second: 2.5 ms for layer 2's chained launch at batch
//     4096 against 2.5 ms for the two kernels it replaces; the steps were issue- and barrier-bound, not HBM-bound.)
//   * A wave's vmcnt counts its LDS-DMA operations and its stores together, and nothing guarantees that the two kinds retire
//     in order with respect to each other (round 2 found that for LDS-DMA and ordinary loads: DESIGN.md section 4.4).  The
//     counted waits are therefore written so that they hold whatever the stores do: a wait names N = the number of YOUNGER
//     LDS-DMA operations only.  If the operation waited for were still outstanding, all N younger ones would be too (LDS-DMA
//     retires in order among itself) -- more than N; so it has landed.  The price is that a wave's stores must have retired as
//     well; the loaders therefore never store in the steady state (the storers write Y), only a tile's t1' once per tile.
//   * The LDS images are rows of 64 bytes; 16-byte slot of k-chunk q in row r: (r >> 2 & 3) ^ H[q],
//     H = {0, 3, 1, 2} -- conflict-free for gfx950's ds_read_b128 lane groups (rows 4 apart share banks), applied on the
//     SOURCE side of the LDS-DMA (the destination is lane-linear).
// Bit-identical to the two-kernel path: same k order, fp32 accumulation, one rounding of Y to fp16 before it is used again.
// NW = waves per workgroup (8 or 16), NSPLIT = waves that share a 16-pixel row tile (1 or 2): a tile is 16 NW / NSPLIT pixels, a
// ring item that many rows x 32 halves.  With NSPLIT = 2 (16 waves on a 128-pixel tile) the two waves of a row tile split the
// channels -- one of the chunk's two 16-channel tiles each in GEMM 3, half of MIDN each in GEMM 1' -- so that the accumulators
// fit 128 registers per lane and four waves per SIMD issue where two did (the 256-channel shapes are issue-bound, not
// memory-bound: DESIGN.md section 4.4b).

#ifdef SALVE_BUILD_ABLATIONS
#define CHAIN_DBG(p, bit) (((p).dbg & (bit)) != 0)
#else
#define CHAIN_DBG(p, bit) false
#endif
struct ChainArgs {
    const uint16_t* t2;    // [M][MID]
    const uint16_t* x;     // [M][4 MID]   residual
    uint16_t* y;           // [M][4 MID]
    uint16_t* t1n;         // [M][MIDN]    (CHAIN)
    const uint16_t* wc;    // [4 MID][MID]
    const uint16_t* wa;    // [MIDN][4 MID] (CHAIN)
    const float* bc;
    const float* ba;
    const uint16_t* zeros;
    int M, n_tiles;
    int y_even_w;          // 0: every row of Y is stored.  W > 0 (r5): the images are W x W and Y's only reader besides the chained
                           // convolution is a stride-2 projection shortcut (in2_buf, stride2 = 2) -- only the pixels with even row AND
                           // even column are ever read, so only those are written: three quarters of Y's bytes never cross HBM
    int32_t* status;
    int dbg;   // ABLATION BUILD ONLY (-DSALVE_BUILD_ABLATIONS; timing only, wrong results): 1 = no weight LDS-DMA after the prologue,
               // 2 = no item LDS-DMA, 4 = no Y stores.  The product build compiles the tests of these bits out (CHAIN_DBG).
};

__device__ __forceinline__ int ec_slot(int row, int q) {   // element offset of k-chunk q of row `row` in a [rows][32] image
    const int h = (0x9C >> (2 * q)) & 3;                   // H = {0, 3, 1, 2}, two bits each
    return row * 32 + ((((row >> 2) & 3) ^ h) << 3);
}
__device__ __forceinline__ int ec_src_chunk(int row, int slot) {   // which k-chunk the lane writing (row, slot) must fetch
    const int v = slot ^ ((row >> 2) & 3);
    return (0x78 >> (2 * v)) & 3;                           // the inverse of H: {0, 2, 3, 1}
}

template <int MID, int MIDN, int R, bool CHAIN, int AHEAD = 2, int NW = 8, int NSPLIT = 1>
__global__ __launch_bounds__(NW * 64, NW / 4) void expand_chain_kernel(ChainArgs p) {
    constexpr int EC_THREADS = NW * 64, ROWS = NW * 16 / NSPLIT, EC_ITEM_E = ROWS * 32;
    constexpr int NL = NW / 2;              // loader waves (0 .. NL - 1); the others are the storers
    constexpr int NJ = 2 / NSPLIT;          // GEMM 3: 16-channel tiles of the 32-channel chunk per wave
    constexpr int C4 = 4 * MID;
    constexpr int KA = MID / 32;            // A items (k-steps of t2) per tile
    constexpr int NCH = C4 / 32;            // RES items (chunks of Y) per tile
    constexpr int SPT = KA + NCH;           // steps per tile
    constexpr int WC_E = 32 * MID;          // one Wc chunk: [KA][32 rows][32]
    constexpr int WA_E = CHAIN ? MIDN * 32 : 0;   // one Wa' chunk: [MIDN rows][32]
    constexpr int NT1 = CHAIN ? MIDN / 16 / NSPLIT : 1;   // GEMM 1': 16-channel tiles per wave
    constexpr int WC_DMA = 2 * KA / NL;     // LDS-DMA instructions per loader wave: Wc chunk (1 KB each) ...
    constexpr int WA_DMA = CHAIN ? MIDN / 16 / NL : 0;  // ... Wa' chunk ...
    constexpr int IT_DMA = (ROWS / 16) / NL;            // ... ring item
    constexpr int PIECES = ROWS * 4 / (NL * 64);        // 16-byte pieces of a Y chunk per storer lane
    static_assert(2 * KA % NL == 0 && (!CHAIN || (MIDN / 16) % NL == 0) && IT_DMA >= 1 && PIECES >= 1 && (NSPLIT == 1 || NW == 16), "work split over the waves");
    constexpr int NB_C = AHEAD + 1, NB_A = AHEAD + 2;   // Wc / Wa' buffers: a chunk is issued AHEAD RES steps ahead into the buffer read last one / two steps ago
    constexpr int SMEM_E = NB_C * WC_E + NB_A * WA_E + R * EC_ITEM_E + 2 * (C4 + (CHAIN ? MIDN : 0));
    static_assert(R >= 4, "an item must be older than the previous step's issue when it is consumed");
    static_assert(AHEAD >= 2, "the wait at the end of a step leaves that step's issues in flight: the next step's weights must be older "
                              "(AHEAD = 1 was tried for a deeper item ring: the bit-identity test caught the race at once)");
    static_assert(SMEM_E * 2 <= 160 * 1024, "LDS budget");
    static_assert(SMEM_E * 2 > 80 * 1024, "one workgroup per CU by construction: keep the LDS image above half a CU's LDS");
    __shared__ __attribute__((aligned(1024))) uint16_t smem[SMEM_E];
    uint16_t* WcB = smem;
    uint16_t* WaB = WcB + NB_C * WC_E;
    uint16_t* Ring = WaB + NB_A * WA_E;
    float* biasc = reinterpret_cast<float*>(Ring + R * EC_ITEM_E);
    float* biasa = biasc + C4;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frag_row = lane & 15, frag_q = lane >> 4;
    const bool loader = wave < NL;
    const int sw = wave % NL;                // index among the loaders / among the storers
    // this workgroup's tiles: b, b + G, b + 2 G, ...
    const int G = gridDim.x, b = blockIdx.x;
    const int n_my = b < p.n_tiles ? (p.n_tiles - b + G - 1) / G : 0;
    if (n_my == 0) return;
    const int T = n_my * SPT;                // steps of this workgroup

    // ---- biases -> LDS (ordinary loads, before any LDS-DMA is in flight)
    for (int i = tid; i < C4; i += EC_THREADS) biasc[i] = p.bc[i];
    if (CHAIN) for (int i = tid; i < MIDN; i += EC_THREADS) biasa[i] = p.ba[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- LDS-DMA issue (loader waves only).  One instruction = 16 rows x 64 bytes; lane l -> row l >> 2, slot l & 3.
    // Addresses are a wave-uniform 64-bit base (scalar arithmetic) plus a 32-bit per-lane byte offset that is computed ONCE
    // (per kernel for the weights, per tile for the items): the first version recomputed 64-bit per-lane addresses, row guards
    // and zero-page selects for every instruction and spent 465 vector + 215 scalar instructions per step -- the step time was
    // its own address arithmetic, not memory (timing-only ablation: all memory operations removed, 78 % of the time stayed).
    const int drow = lane >> 2, dslot = lane & 3;
    uint32_t wc_off[WC_DMA > 0 ? WC_DMA : 1], wa_off[WA_DMA > 0 ? WA_DMA : 1];
#pragma unroll
    for (int u = 0; u < WC_DMA; u++) {
        const int idx = sw * WC_DMA + u, ks = idx >> 1, row = (idx & 1) * 16 + drow;   // k-step idx >> 1, row half idx & 1
        wc_off[u] = (uint32_t)((row * MID + ks * 32 + ec_src_chunk(row, dslot) * 8) * 2);
    }
#pragma unroll
    for (int u = 0; u < WA_DMA; u++) {
        const int row = (sw * WA_DMA + u) * 16 + drow;
        wa_off[u] = (uint32_t)((row * C4 + ec_src_chunk(row, dslot) * 8) * 2);
    }
    // producer cursor of the item stream: next item = tile p_ti (of this workgroup), step p_s, into ring slot p_slot
    int p_g = 0, p_ti = 0, p_s = 0, p_slot = 0;
    uint32_t it_off_a[IT_DMA], it_off_x[IT_DMA];   // per-lane byte offsets into the producer's tile (rows beyond M: the last row)
    const char* it_base_a = nullptr;               // wave-uniform: t2 / x at the producer's tile
    const char* it_base_x = nullptr;
    auto producer_tile = [&]() {
        const long long m0 = ((long long)b + (long long)p_ti * G) * ROWS;
        const long long left = (long long)p.M - m0;
        const int rows = left >= ROWS ? ROWS : (left > 0 ? (int)left : 1);
        const long long mb = left > 0 ? m0 : 0;    // (a tile past the end of the stream: any valid rows, nobody reads the slot)
        it_base_a = reinterpret_cast<const char*>(p.t2) + mb * (MID * 2);
        it_base_x = reinterpret_cast<const char*>(p.x) + mb * (C4 * 2);
#pragma unroll
        for (int u = 0; u < IT_DMA; u++) {
            const int row = (sw * IT_DMA + u) * 16 + drow;
            const int rc = row < rows ? row : rows - 1;   // rows beyond M read the tile's last row: never stored, never read by a kept pixel
            const int ch = ec_src_chunk(row, dslot) * 8;
            it_off_a[u] = (uint32_t)((rc * MID + ch) * 2);
            it_off_x[u] = (uint32_t)((rc * C4 + ch) * 2);
        }
    };
    auto issue_item = [&]() {
        const bool isA = p_s < KA;
        const char* base = isA ? it_base_a + p_s * 64 : it_base_x + (p_s - KA) * 64;
        uint16_t* dst = Ring + p_slot * EC_ITEM_E + sw * IT_DMA * 512;
#pragma unroll
        for (int u = 0; u < IT_DMA; u++)
            __builtin_amdgcn_global_load_lds((global_cptr)(base + (isA ? it_off_a[u] : it_off_x[u])), (lds_ptr)(dst + u * 512), 16, 0, 0);
        p_g++;
        if (++p_s == SPT) { p_s = 0; p_ti++; producer_tile(); }
        if (++p_slot == R) p_slot = 0;
    };
    // weight streams: chunk w_nc goes to Wc buffer w_cb and Wa' buffer w_ab, AHEAD RES steps ahead of its use
    int w_nc = 0, w_cb = 0, w_ab = 0;
    auto issue_weights = [&]() {
        {
            const char* base = reinterpret_cast<const char*>(p.wc) + (long long)w_nc * (32 * MID * 2);
            uint16_t* dst = WcB + w_cb * WC_E + sw * WC_DMA * 512;
#pragma unroll
            for (int u = 0; u < WC_DMA; u++) __builtin_amdgcn_global_load_lds((global_cptr)(base + wc_off[u]), (lds_ptr)(dst + u * 512), 16, 0, 0);
        }
        if constexpr (CHAIN) {
            const char* base = reinterpret_cast<const char*>(p.wa) + w_nc * 64;
            uint16_t* dst = WaB + w_ab * WA_E + sw * WA_DMA * 512;
#pragma unroll
            for (int u = 0; u < WA_DMA; u++) __builtin_amdgcn_global_load_lds((global_cptr)(base + wa_off[u]), (lds_ptr)(dst + u * 512), 16, 0, 0);
        }
        if (++w_nc == NCH) w_nc = 0;
        if (++w_cb == NB_C) w_cb = 0;
        if (++w_ab == NB_A) w_ab = 0;
    };

    // ---- prologue: the first R - 2 items, AHEAD chunks of each weight stream; one full wait (once per workgroup)
    if (loader) {
        producer_tile();
        for (int g = 0; g < R - 2; g++) issue_item();
        for (int a = 0; a < AHEAD; a++) issue_weights();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");

    act8 afr[KA];                      // this wave's A fragments: pixels 16 wave + frag_row, all of K
    f32x4 acc1[NT1];                   // this wave's t1' accumulators: the same 16 pixels, all MIDN channels
#pragma unroll
    for (int k = 0; k < KA; k++) afr[k] = act8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int n = 0; n < NT1; n++) acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float amax = 0.f;
    const int mt = wave % (NW / NSPLIT), np_ = wave / (NW / NSPLIT);   // this wave's 16-pixel row tile, and which part of the channels
    const int my_row = mt * 16 + frag_row;         // row of the tile this lane's outputs belong to
    // per-lane constants of the LDS accesses (element offsets inside an item / a weight buffer)
    const int a_frag_off = ec_slot(my_row, frag_q);                 // this lane's A fragment in an item (t2 k-step or Y chunk)
    const int b_frag_off = ec_slot(frag_row, frag_q);               // B fragment of a 16-row tile (+ 512 elements per tile)
    int cell_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) cell_off[j] = ec_slot(my_row, (16 * (j + np_ * NJ) + 4 * frag_q) >> 3) + ((16 * (j + np_ * NJ) + 4 * frag_q) & 7);
    int piece_off[PIECES];                                          // storer lanes: 16-byte pieces of a Y chunk
    uint32_t piece_goff[PIECES];                                    // ... and their byte offsets in Y relative to the tile's chunk
#pragma unroll
    for (int it = 0; it < PIECES; it++) {
        const int piece = it * (NL * 64) + sw * 64 + lane, row = piece >> 2, q = piece & 3;
        piece_off[it] = ec_slot(row, q);
        piece_goff[it] = (uint32_t)((row * C4 + q * 8) * 2);
    }
    int c_slot = 0, c_cb = 0, c_ab = 0;            // consumer cursor: ring slot, weight buffers of the current RES step
    int prev_slot = 0, prev_ab = 0, prev_nc = 0;   // the previous RES step's Y chunk (GEMM 1' and the store run one step late)
    long long prev_m0 = 0;
    constexpr bool YEVEN = CHAIN && MIDN != MID;   // only a stage's last block can have y_even_w set (the host checks): the others carry no mask
    static_assert(PIECES <= 16, "two keep masks in one register");
    uint32_t keeps = ~0u;                          // y_even_w: bit `it` (16 + `it`) = this lane's piece `it` of the current (previous) tile belongs to a stored pixel
    auto tile_keep = [&](long long m0_) -> uint32_t {
        if (!YEVEN || p.y_even_w == 0) return ~0u;
        uint32_t k = 0;
#pragma unroll
        for (int it = 0; it < PIECES; it++) {
            const long long m = m0_ + ((it * (NL * 64) + sw * 64 + lane) >> 2);
            const int ox = (int)(m % p.y_even_w), oy = (int)((m / p.y_even_w) % p.y_even_w);
            k |= (((ox | oy) & 1) == 0 ? 1u : 0u) << it;
        }
        return k;
    };
    bool tail = false;                             // the producer has run past the end of the stream: waits are full from here on

    // GEMM 1' of the previous step's Y chunk, and the tile's t1' when that chunk was its last
    auto chain_prev = [&]() {
        if constexpr (CHAIN) {
            const uint16_t* Yc = Ring + prev_slot * EC_ITEM_E;
            const uint16_t* Wn = WaB + prev_ab * WA_E + b_frag_off + np_ * NT1 * 512;
            // all fragment reads of a batch first, then its MFMAs: left to itself hipcc re-uses one register quad for every
            // B fragment and emits read -> wait -> MFMA per tile, i.e. an LDS round trip in front of every MFMA
            const act8 ay = *reinterpret_cast<const act8*>(Yc + a_frag_off);
            constexpr int NB = NW == 16 ? 4 : (NT1 < 16 ? NT1 : 16);   // (16 waves: 128 registers per lane -- smaller batches)
#pragma unroll
            for (int n0 = 0; n0 < NT1; n0 += NB) {
                act8 bn[NB];
#pragma unroll
                for (int n = 0; n < NB; n++) bn[n] = *reinterpret_cast<const act8*>(Wn + (n0 + n) * 512);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NB; n++) acc1[n0 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bn[n], ay, acc1[n0 + n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (prev_nc == NCH - 1) {
                // bias + ReLU, one rounding, straight from the accumulators: 8 bytes per lane, a wave-instruction writes 16 rows
                // x 32 contiguous bytes.  (The only stores a loader wave ever issues: once per tile.)
                const long long m = prev_m0 + my_row;
                float4 bias[NT1];
#pragma unroll
                for (int n = 0; n < NT1; n++) bias[n] = lds_read_f4(biasa + 16 * (n + np_ * NT1) + 4 * frag_q);
                uint2 o[NT1];
#pragma unroll
                for (int n = 0; n < NT1; n++) {
                    o[n] = pack4<true>(amax, acc1[n] + vec4(bias[n]));
                    acc1[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (m < p.M) {
                    uint16_t* dst = p.t1n + m * MIDN + 16 * np_ * NT1 + 4 * frag_q;
#pragma unroll
                    for (int n = 0; n < NT1; n++) *reinterpret_cast<uint2*>(dst + 16 * n) = o[n];
                }
            }
        }
    };
    auto store_prev = [&]() {   // storer waves: the previous RES step's Y chunk, 512 pieces of 16 bytes, 2 per lane (a row's 64 bytes = 4 lanes)
        if (CHAIN_DBG(p, 4)) return;
        const uint16_t* Yc = Ring + prev_slot * EC_ITEM_E;
        char* ybase = reinterpret_cast<char*>(p.y) + prev_m0 * (C4 * 2) + prev_nc * 64;   // wave-uniform
        act8 vh[PIECES];
#pragma unroll
        for (int it = 0; it < PIECES; it++) vh[it] = *reinterpret_cast<const act8*>(Yc + piece_off[it]);
        if constexpr (PIECES == 2) asm volatile("" : "+v"(vh[0]), "+v"(vh[1]));
        else asm volatile("" : "+v"(vh[0]));
        uint4 v[PIECES];
#pragma unroll
        for (int it = 0; it < PIECES; it++) v[it] = __builtin_bit_cast(uint4, vh[it]);
        if (prev_m0 + ROWS <= p.M && (!YEVEN || p.y_even_w == 0)) {       // a full tile, every row stored (wave-uniform): no row guards
#pragma unroll
            for (int it = 0; it < PIECES; it++) *reinterpret_cast<uint4*>(ybase + piece_goff[it]) = v[it];
        } else {
#pragma unroll
            for (int it = 0; it < PIECES; it++) {
                const int row = (it * (NL * 64) + sw * 64 + lane) >> 2;
                if (prev_m0 + row < p.M && (!YEVEN || ((keeps >> (16 + it)) & 1u))) *reinterpret_cast<uint4*>(ybase + piece_goff[it]) = v[it];
            }
        }
    };
    // the loader's side of a step: issue for later steps; returns nothing, records whether the stream has ended
    auto loader_issue = [&](bool res_step) {
        if (p_g < T) { if (!CHAIN_DBG(p, 2)) issue_item(); else { p_g++; if (++p_s == SPT) { p_s = 0; p_ti++; } if (++p_slot == R) p_slot = 0; } }
        else tail = true;
        if (res_step && !CHAIN_DBG(p, 1)) issue_weights();
    };

    for (int ti = 0; ti < n_my; ti++) {
        const long long m0 = ((long long)b + (long long)ti * G) * ROWS;
        if (YEVEN && !loader) keeps = (keeps & 0xffff0000u) | (tile_keep(m0) & 0xffffu);
        // ---------------------------------------------------------------- A items: the tile's A fragments -> registers
#pragma unroll
        for (int k = 0; k < KA; k++) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");   // the item is visible (the loaders waited for it at the end of the previous
                                                      // step); every LDS access of the previous step is done
            if (loader) loader_issue(false);          // -> the slot read last two steps ago
            if (k == 0 && ti > 0) {                   // the previous tile's last chunk: store, GEMM 1', t1'
                if (!loader) store_prev();
                chain_prev();
            }
            afr[k] = *reinterpret_cast<const act8*>(Ring + c_slot * EC_ITEM_E + a_frag_off);
            if (loader) {   // the NEXT step's data has landed: everything but what this step's issue put in flight
                if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT_DMA) : "memory");
            }
            if (++c_slot == R) c_slot = 0;
        }
        // ---------------------------------------------------------------- RES items: chunk nc of Y
        for (int nc = 0; nc < NCH; nc++) {
            uint16_t* slot = Ring + c_slot * EC_ITEM_E;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (loader) loader_issue(true);           // item -> the slot read last two steps ago; Wc -> the buffer GEMM 3 read one
                                                      // RES step ago; Wa' -> the one GEMM 1' read then
            if (nc > 0) {
                if (!loader) store_prev();
                chain_prev();
            }
            const uint16_t* Wb = WcB + c_cb * WC_E + b_frag_off + np_ * NJ * 512;
            f32x4 acc3[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) acc3[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the epilogue's LDS reads (bias, residual cell) are issued in front of GEMM 3's: LDS returns in order, so they ride
            // under the multiply instead of costing a round trip of their own after it
            act8 bias_h[NJ];                       // (raw _Float16-typed loads; laundered together below: one asm, one wait)
            half4v rr_h[NJ];
            constexpr int KB = NW == 16 ? 2 : (KA < 8 ? KA : 8);   // k-steps per batch of fragment reads (16 reads = 64 VGPRs)
#pragma unroll
            for (int k0 = 0; k0 < KA; k0 += KB) {
                act8 bfr[KB][NJ];
#pragma unroll
                for (int ks = 0; ks < KB; ks++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) bfr[ks][j] = *reinterpret_cast<const act8*>(Wb + (k0 + ks) * 1024 + j * 512);
                if (k0 + KB >= KA) {
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        bias_h[j] = *reinterpret_cast<const act8*>(biasc + nc * 32 + 16 * (j + np_ * NJ) + 4 * frag_q);
                        rr_h[j] = *reinterpret_cast<const half4v*>(slot + cell_off[j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KB; ks++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfr[ks][j], afr[k0 + ks], acc3[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bias + residual + ReLU on the accumulator's own elements, in place in the ring slot (fp32, one rounding)
            if constexpr (NJ == 2) asm volatile("" : "+v"(bias_h[0]), "+v"(bias_h[1]), "+v"(rr_h[0]), "+v"(rr_h[1]));   // keeps the loads <n x half>-typed (lds_read8)
            else asm volatile("" : "+v"(bias_h[0]), "+v"(rr_h[0]));
            float4 bias[NJ];
            uint2 rr[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) { bias[j] = __builtin_bit_cast(float4, bias_h[j]); rr[j] = __builtin_bit_cast(uint2, rr_h[j]); }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                lds_write8(slot + cell_off[j], pack4<true>(amax, acc3[j] + vec4(bias[j]) + vec4(rr[j])));
            }
            if (loader) {   // the NEXT step's data has landed: everything but what this step's issue put in flight
                if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT_DMA + WC_DMA + WA_DMA) : "memory");
            }
            prev_slot = c_slot; prev_ab = c_ab; prev_nc = nc; prev_m0 = m0; if constexpr (YEVEN) keeps = (keeps & 0xffffu) * 0x10001u;
            if (++c_cb == NB_C) c_cb = 0;
            if (++c_ab == NB_A) c_ab = 0;
            if (++c_slot == R) c_slot = 0;
        }
    }
    // the last chunk: its store, its GEMM 1' and the last tile's t1' (slot and Wa' buffer are intact: the issues of the last two
    // steps went to older buffers)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (!loader) store_prev();
    chain_prev();
    // nothing may still be landing in LDS when the workgroup leaves
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    report_range(p.status, amax);
}
