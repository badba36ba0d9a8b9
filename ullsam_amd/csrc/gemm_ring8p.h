// The ring kernel made PERSISTENT (round 6).  Included by gemm.hip after gemm_ring8_kernel (same helpers: GemmArgs, glds16_sbase, pair_swap,
// pack_bf16x2); same tile shapes, same LDS image, same load / matrix slots, same direct epilogues -- what changes is the OUTER structure:
//
//   * a grid of at most one workgroup per CU; workgroup b walks the tiles b, b + G, b + 2 G, ... (G = gridDim.x) in the dispatch order the
//     one-tile-per-workgroup kernel has, so the sets of tiles that run together (and share an XCD's L2) are the same;
//   * the LDS ring NEVER drains at a tile border: the last two load slots of a tile request stages 0 and 1 of the NEXT tile (in the one-tile kernel they request
//     nothing); in the DEFAULT schedule (SCHED 2) that is all -- every trip of the K loop is the same code, the next tile's first load slot requests its stage 2 as
//     every load slot requests the stage two ahead; the first schedule (SCHED 0) also requested stage 2 at the border and had a request-free first stage, which cost it
//     peeled first / last trips;
//   * loads and stores share `vmcnt` on gfx950 and complete in issue order, which is what sank the persistent kernels of rounds 2 and 3 (the next tile's counted
//     waits sat behind the previous tile's 16 - 40 store acknowledgements).  Here the border ends with ONE `s_waitcnt vmcnt(0)` -- stages 0' and 1' have long landed,
//     only the youngest stores are still on their way (330 - 670 cycles by the stamps) -- and no counted wait of the next tile has a store in front of it;
//   * the two wave groups keep their one-slot stagger inside a tile; at the border (SCHED 2) group 0 waits one extra barrier for group 1's last matrix slot, both
//     groups convert and store their accumulators AT THE SAME TIME (as in the one-tile kernel), and group 1 takes one extra barrier behind its vmcnt(0) to fall one
//     slot behind again (SCHED 0 ran the two epilogues one after the other and so gave back the prologue it saved);
//   * requests go through BUFFER descriptors (`buffer_load_dwordx4 ... offen lds`: descriptor + per-lane offset + scalar offset): the per-lane
//     offsets are the same in every tile, a tile switch is one scalar per operand, and rows past M are out of the descriptor's range (no fetch, no
//     clamp) -- so ONE copy of the load slot serves every stage of every tile, with the request's scalar offset switched two stages before a border;
//   * the bias row of the NEXT tile travels by LDS-DMA into the other of two 2 KiB buffers behind the ring, requested by waves 0 / 1 after
//     their stores (group 1 may still be reading the current one).
// What it removes per tile after a workgroup's first: the prologue (two stages requested and the first awaited with nothing else to do:
// 4.2 - 5.3 k cycles), the workgroup launch and the wait for the last stores before a wave may end (wave lifetime minus stamped span: ~9 k
// cycles on llm.w13), and the dispatch gap between rounds.  Requirements (checked by the launcher, everything else goes to gemm_ring8_kernel):
// bf16, N % BN == 0, K % 128 == 0 (the ring slot of a stage is its index & 3 in every tile), a direct epilogue (16-byte aligned rows, even M),
// operands below 4 GiB (32-bit descriptor offsets).
//
// Compiler notes (hipcc, ROCm 7.2), each found in the first build of this file:
//   * everything the epilogue derives from the lane id is invariant in the tile loop and LICM hoists it across the K loop, where the registers do
//     not exist -> an opaque copy of the lane id per trip;
//   * a value reloaded from scratch (or loaded from memory) in front of the K loop and first used inside it puts a compiler `s_waitcnt vmcnt(0)`
//     INSIDE the load slot (the wait-count pass cannot hoist it out of the loop), draining the ring every stage -> the loop's per-lane inputs are
//     "touched" by an empty asm in front of the loop, so any such wait lands there;
//   * a wave-uniform but run-time sub-tile row count (9 / 8 rows in the two wave rows of the 272-row tile) makes every copy of the load and matrix
//     slots carry branches and undefined-value PHIs that the allocator spills -> the two wave rows run separate instantiations of the tile loop.

// One LDS-DMA request through a buffer descriptor: 16 bytes per lane from desc.base + voff (per lane) + soff (wave-uniform) -> LDS at lds_addr + 16 lane.
__device__ __forceinline__ void blds16(u32x4 desc, unsigned int voff, unsigned int soff, unsigned int lds_addr) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(desc), "s"(soff), "s"(lds_addr) : "memory");
}
// The same with the LDS target as (wave base in an SGPR) + a compile-time offset: M0 is written by the add itself and NOT restored (nothing else in the
// persistent kernel uses M0: its other LDS-DMA, the bias row, saves and restores it).  (The instruction's immediate offset is not used for the k offset:
// it advances the LDS address as well as the memory address.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int MOFF>
__device__ __forceinline__ void blds16_imm(u32x4 desc, unsigned int voff, unsigned int soff, unsigned int wave_lds_base) {
    asm volatile("s_add_i32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(desc), "s"(soff), "s"(wave_lds_base), "n"(MOFF) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 12, "counted waits of the ring");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
#pragma clang diagnostic pop
// The lane id, re-derived from the hardware (two VALU instructions) at every use site of the persistent kernel's tile loop: a lane id kept in a register
// across the loop's phases is one more value for the allocator to spill across the epilogue and reload -- with a compiler `s_waitcnt vmcnt(0)` -- in front
// of the K loop; as volatile asm it is also opaque to LICM (what is derived from it stays inside the trip).
__device__ __forceinline__ int fresh_lane() {
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
}
__device__ __forceinline__ u32x4 raw_desc(const void* base, unsigned int bytes) {
    const uintptr_t b = reinterpret_cast<uintptr_t>(base);
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((unsigned int)b);
    d[1] = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32) & 0xffffu);   // stride 0: raw buffer, range check against num_records bytes
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}

// DIST: stages between a request and its use in the two-barrier schedule (2: the one-tile kernel's; 3: an experiment).
// SCHED 0: the one-tile kernel's schedule -- two barriers per stage, the groups' load and matrix slots strictly alternating.
// SCHED 2: SCHED 0's stage (two barriers), but (a) every stage of every trip requests (the border requests nothing: the next tile's first load slot requests its stage 2 behind the border's
//   vmcnt(0)), so the K loop is ONE loop of identical trips -- the last trip's stages 2 / 3 take the next tile's scalar offsets through two selects per TRIP -- with no peeled copies (the peeled
//   first / last trips of SCHED 0 are where the 256x320 instantiation spilled); (b) the two wave groups run their epilogues AT THE SAME TIME, as in the one-tile kernel: group 0 waits one
//   barrier at the border's start (for group 1's last matrix slot), group 1 one at its end (for group 0's first load slot) -- SCHED 0 ran them one after the other and gave back the prologue it saved.
// SCHED 1: ONE barrier per stage.  Between two barriers group 0 runs [matrix slot of stage s, load slot of stage s + 1] and group 1 [load slot of stage s, matrix slot
//   of stage s, counted wait]: the matrix pipe always has one group's instructions to run and a barrier's latency is paid once per stage, not twice (in-stage
//   stamps of SCHED 0: a load slot's own work ends ~340 cycles into a slot of ~680 whose length is the partner's 512 - 576 matrix cycles plus ~120 cycles around the
//   two barriers).  Safety with the four-slot ring: a stage is read by group 0 in the interval BEFORE the barrier after which group 1 reads it, so group 1's pieces
//   must have landed one barrier earlier than in SCHED 0 -- group 1 requests THREE stages ahead (its slot was last read, by itself, two intervals earlier, and by
//   group 0 three), group 0 two; every wave's counted wait still leaves exactly one stage of its own pieces in flight.
// ABL (diagnostic, wrong results): 1 = the K loop issues no LDS-DMA request, 2 = no fragment read (the MFMAs run on whatever the registers hold), 3 = neither -- what a
// stage costs without each ingredient (tools/persist_ab.py modes 5 / 6 / 7).
template <int MI0, int MI1, int NTW, int EMODE = 0, bool STAMP = false, int DIST = 2, int SCHED = 0, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_ring8p_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef bf16 T;
    constexpr int BM = 16 * (MI0 + MI1), BN = 64 * NTW;
    constexpr int PA = BM / 16, PB = BN / 16;          // DMA pieces (16 rows x 64 B) per stage
    constexpr int ASZ = BM * 64, STG = ASZ + BN * 64;
    constexpr int BIAS_LDS = 4 * STG;                  // two bias rows (2 KiB each) behind the ring
    static_assert(PA >= 8 && PA <= 24 && PB >= 8 && PB <= 24, "piece assignment below: one to three pieces per wave and operand");
    static_assert(NTW == 4 || NTW == 5, "epilogue layouts below: four sub-tiles in two pairs, optionally a fifth on its own");
    constexpr int WW = 16 * NTW;

    const int nblk = p.full_tiles;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int GMr = p.group_m;
    const int width = GMr * p.tiles_n;
    auto tile_of = [&](int v, int& tm, int& tn) __attribute__((always_inline)) {   // virtual block index -> tile (the one-tile kernel's XCD + raster mapping)
        const int xcd = v & 7;
        const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        const int group = swz / width;
        const int first_m = group * GMr;
        const int gsize = min(p.tiles_m - first_m, GMr);
        const int rem = swz - group * width;
        tn = rem / gsize;
        tm = first_m + (rem - tn * gsize);
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, grp = wave >> 2;
    const int na = (PA - wave + 7) / 8, nb = (PB - wave + 7) / 8;
    const int perm = EMODE == 1 ? 2 : p.out_f32 ? 0 : (p.act == 3 ? 2 : 1);
    auto w_row = [&](int R) {
        const int wb = R / WW, q = R - wb * WW, j = q >> 4, r = q & 15;
        if (perm == 1) return j < 4 ? WW * wb + 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3) : WW * wb + 64 + r;
        if (perm == 2) return 128 * (wb >> 1) + 64 * (j >> 1) + 32 * (wb & 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3);
        return R;
    };
    // per-lane source offsets inside a tile (the same in every tile: rows past M are out of the descriptor's range, N % BN == 0).  They -- and the LDS read
    // bases below -- are RECOMPUTED from an opaque copy of the lane id at the start of every tile (a few dozen VALU instructions per tile) instead of living
    // across the epilogue, whose registers they would otherwise take: the 256x320 instantiation spilled 80 registers with them live.
    unsigned int a_off[3], b_off[3];
    auto set_offsets = [&](int ln) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int row = (wave + 8 * i) * 16 + (ln >> 2);
            const int c = (ln & 3) ^ ((row >> 2) & 2);
            a_off[i] = (unsigned int)row * (unsigned int)(p.lda * 2) + (c << 4);
            b_off[i] = (unsigned int)w_row(row) * (unsigned int)(p.ldw * 2) + (c << 4);
        }
    };
    set_offsets(lane);
    const u32x4 a_desc = raw_desc(p.A, (unsigned int)(((size_t)(p.M - 1) * p.lda + p.K) * 2));
    const u32x4 b_desc = raw_desc(p.W, (unsigned int)(((size_t)(p.N - 1) * p.ldw + p.K) * 2));
    const unsigned int a_tile = (unsigned int)(p.lda * 2) * BM, b_tile = (unsigned int)(p.ldw * 2) * BN;   // bytes from one tile row / column to the next
    const int st1 = p.K >> 5;  // stages (32-deep k-steps) per tile, a multiple of 4, at least 12
    const unsigned int wbase = __builtin_amdgcn_readfirstlane((unsigned int)(uintptr_t)LDS_PTR(smem + wave * 1024));   // LDS address of this wave's first piece in ring slot 0
    const bool has_bias = p.bias && p.act != 3 && EMODE == 0;
    const int G = gridDim.x;

    auto bias_dma = [&](int n0t, int buf) __attribute__((always_inline)) {   // waves 0 / 1: the tile's bias row -> bias buffer `buf`
        // (as asm: an LDS-DMA issued through the builtin is a pending LDS write in hipcc's wait-count model)
        if (has_bias && wave < 2) {
            const int c = min(n0t + wave * 256 + fresh_lane() * 4, p.N - 4);
            glds16_sbase(reinterpret_cast<const char*>(p.bias), (unsigned int)c * 4u, (unsigned int)(uintptr_t)LDS_PTR(smem + BIAS_LDS + buf * 2048 + wave * 1024));
        }
    };
    auto request_stage = [&](unsigned int ra, unsigned int rb, int slot) __attribute__((always_inline)) {   // all of this wave's pieces of one stage (prologue, tile border)
        char* base = smem + slot * STG;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (i < na) blds16(a_desc, a_off[i], ra, (unsigned int)(uintptr_t)LDS_PTR(base + (wave + 8 * i) * 1024));
            if (i < nb) blds16(b_desc, b_off[i], rb, (unsigned int)(uintptr_t)LDS_PTR(base + ASZ + (wave + 8 * i) * 1024));
        }
    };
    auto stamp = [&](int tile_k, int kk) __attribute__((always_inline)) {   // (diagnostic build) [workgroup][group][tile < 8][4]
        if (STAMP && p.dbg && (tid & 255) == 0 && tile_k < 8) p.dbg[(((size_t)blockIdx.x * 2 + grp) * 8 + tile_k) * 4 + kk] = __builtin_amdgcn_s_memtime();
    };

    // ---- first tile: bias row and stages 0, 1, 2 requested and landed (every tile then starts from the same state: stage 0 requests nothing)
    int tm0, tn0;
    tile_of(blockIdx.x, tm0, tn0);
    bias_dma(tn0 * BN, 0);
    request_stage(tm0 * a_tile, tn0 * b_tile, 0);
    request_stage(tm0 * a_tile + 64, tn0 * b_tile + 64, 1);
    if (SCHED == 0 || (SCHED == 1 && grp == 1)) request_stage(tm0 * a_tile + 128, tn0 * b_tile + 128, 2);
    if constexpr (SCHED == 0 && DIST >= 3) request_stage(tm0 * a_tile + 192, tn0 * b_tile + 192, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    // ---- everything from here on per wave ROW (MIW sub-tile rows: compile-time) -- the tile loop, its K loop, the epilogue
    auto core = [&](auto MIW_c, auto ROWW_c, auto GRP_c) __attribute__((always_inline)) {
        constexpr int MIW = decltype(MIW_c)::value, GRP = decltype(GRP_c)::value;   // GRP: the wave group (SCHED 1 only; -1: both groups run this instantiation)
        constexpr int DG = SCHED == 1 ? (GRP == 0 ? 2 : 3) : (SCHED == 2 ? 2 : DIST);                  // request distance of this group
        const int row_w = decltype(ROWW_c)::value >= 0 ? decltype(ROWW_c)::value : wm * (MI0 * 16);   // first tile row of this wave's row (run time when both rows share one instantiation)
        f32x4 acc[MIW][NTW];
#pragma unroll
        for (int i = 0; i < MIW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // per-lane LDS read bases of this wave in ring slot 0: row mm of the wave's first A / W sub-tile, 16-byte chunk g4 ^ ((mm >> 2) & 2) (sub-tile rows
        // start at multiples of 16, so the image's chunk swizzle depends on the lane only); sub-tile r is 1 KiB further.  Set per tile (set_lane_state).
        unsigned int rd_a = 0, rd_b = 0;
        auto set_lane_state = [&]() __attribute__((always_inline)) {
            const int ln = fresh_lane();
            set_offsets(ln);
            const unsigned int lane_part = (unsigned int)((ln & 15) * 64 + (((ln >> 4) ^ (((ln & 15) >> 2) & 2)) << 4));
            rd_a = (unsigned int)(uintptr_t)LDS_PTR(smem) + (unsigned int)row_w * 64u + lane_part;
            rd_b = (unsigned int)(uintptr_t)LDS_PTR(smem) + (unsigned int)(ASZ + wn * (16 * NTW) * 64) + lane_part;
        };
        // (diagnostic build) shader-clock stamps INSIDE one stage (tile 1, trip 4, stage 1 of the trip) of every wave: [0] slot start, [1] requests and reads issued,
        // [2] counted wait passed, [3] first barrier passed, [4] matrix instructions issued, [5] second barrier passed -> p.dbg[2^19 + (workgroup x 8 + wave) x 8 + k]
        bool stamp_on = false;
        unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
        // One stage: load slot (this wave's requests for the stage two ahead, alternated with the fragment reads of the current stage, then the counted
        // wait), barrier, matrix slot, barrier.  The K loop is unrolled by the ring's four slots, so that the slot of a stage (read slot J, request slot
        // (J + 2) & 3) and with it every LDS address -- the M0 value of a request, the offsets of the reads -- is a compile-time constant: a request is
        // THREE instructions (s_add_i32 m0, base, imm / s_nop / buffer_load ... offset:imm lds) where the one-tile kernel spends six plus the slot arithmetic
        // (its load slot is the critical path of a stage and carries ~35 scalar instructions per stage; here ~10).  The k offset of a request rides in the
        // instruction's immediate (64 bytes per stage, relative to the trip's scalar offset), so the scalar offsets move once per trip.
        // KIND 0: requests nothing (stage 0 of a tile: its stage 2 went out at the border), 1: requests at so_a / so_b + 64 (J + 2) (this tile),
        // 2: requests at so_a / so_b + 64 (J - 2) (stages 0 / 1 of the NEXT tile: so_* are its offsets).
        auto stage = [&](auto J_c, auto KIND_c, unsigned int so_a, unsigned int so_b, auto NA_c, auto NB_c) __attribute__((always_inline)) {
            constexpr int J = decltype(J_c)::value, KIND = decltype(KIND_c)::value;
            constexpr int NA = decltype(NA_c)::value, NB = decltype(NB_c)::value;
            constexpr int RS = (J + DG) & 3, KOFF = KIND == 2 ? 64 * (J - (4 - DG)) : 64 * (J + DG);
            static_assert(KIND != 2 || J >= 4 - DG, "only the last DG stages of a tile request the next tile");
            static_assert(SCHED == 0 || KIND != 0, "SCHED 1 / 2: every stage requests");
            const unsigned int rq_a = so_a + KOFF, rq_b = so_b + KOFF;   // (two scalar adds per stage)
            // this stage's read bases: the wave's per-lane bases (rd_a / rd_b, below) + the slot's offset, ONE add each, pinned here -- left to itself
            // hipcc keeps a base register per slot and (where it cannot see that the swizzle depends on the lane only) per fragment: 20 registers and
            // spills in the 256x320 instantiation; the fragments then sit at immediate offsets of 1 KiB
            unsigned int t_a = rd_a, t_b = rd_b;
            if constexpr (J != 0) {
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(t_a) : "n"(J * STG), "v"(rd_a));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(t_b) : "n"(J * STG), "v"(rd_b));
            }
            auto frag_at = [&](unsigned int addr) -> Frag<T> {
                Frag<T> f;
                f.v = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8_t*>(addr);
                return f;
            };
            Frag<T> a8[MIW], b[NTW];
            auto tstamp = [&](int k) __attribute__((always_inline)) {
                if constexpr (STAMP && SCHED != 2 && J == 1 && KIND == 1) { if (stamp_on) ts[k] = __builtin_amdgcn_s_memtime(); }
            };
            tstamp(0);
            {
                auto request = [&](auto QI_c) __attribute__((always_inline)) {   // request QI of this wave: A0 B0 A1 B1 A2 B2
                    constexpr int qi = decltype(QI_c)::value, i = qi >> 1;
                    if constexpr (!(ABL & 1) && KIND != 0 && !(qi & 1) && i < NA) blds16_imm<RS * STG + i * 8192>(a_desc, a_off[i], rq_a, wbase);
                    if constexpr (!(ABL & 1) && KIND != 0 && (qi & 1) && i < NB) blds16_imm<RS * STG + ASZ + i * 8192>(b_desc, b_off[i], rq_b, wbase);
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto rd = [&](auto R_c) __attribute__((always_inline)) {
                    constexpr int r = decltype(R_c)::value;
                    if constexpr (r < NTW + MIW) {
                        if constexpr (ABL & 2) {   // no read: an opaque "definition" so that the MFMAs keep their operands
                            if constexpr (r < NTW) asm volatile("" : "=v"(b[r].v)); else asm volatile("" : "=v"(a8[r - NTW].v));
                        } else {
                            if constexpr (r < NTW) b[r] = frag_at(t_b + r * 1024);
                            else a8[r - NTW] = frag_at(t_a + (r - NTW) * 1024);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                // requests and reads alternated: one request, then three reads, ... (the order round 2 measured best)
                request(std::integral_constant<int, 0>{});
                rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{});
                request(std::integral_constant<int, 1>{});
                rd(std::integral_constant<int, 3>{}); rd(std::integral_constant<int, 4>{}); rd(std::integral_constant<int, 5>{});
                request(std::integral_constant<int, 2>{});
                rd(std::integral_constant<int, 6>{}); rd(std::integral_constant<int, 7>{}); rd(std::integral_constant<int, 8>{});
                request(std::integral_constant<int, 3>{});
                rd(std::integral_constant<int, 9>{}); rd(std::integral_constant<int, 10>{}); rd(std::integral_constant<int, 11>{});
                request(std::integral_constant<int, 4>{});
                rd(std::integral_constant<int, 12>{}); rd(std::integral_constant<int, 13>{});
                request(std::integral_constant<int, 5>{});
            }
            tstamp(1);
            auto counted_wait = [&]() __attribute__((always_inline)) {
                if constexpr (KIND == 0) { /* nothing of this wave is in flight (the border's vmcnt(0)) */ }
                else if constexpr (ABL & 1) { /* nothing was requested */ }
                else if constexpr (SCHED == 1 || SCHED == 2) wait_vmcnt<NA + NB>();
                else wait_vmcnt<(DIST - 1) * (NA + NB)>();   // the stage read NEXT has landed: only the youngest stages' pieces may be in flight
            };
            auto matrix_slot = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MIW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) mma16(b[j], a8[i], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            };
            if constexpr (SCHED == 0 || SCHED == 2) {
                counted_wait();
                tstamp(2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                tstamp(3);
                matrix_slot();
                tstamp(4);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                tstamp(5);
            } else if constexpr (GRP == 0) {   // load slot, barrier, matrix slot (the next stage's load slot follows without a barrier)
                counted_wait();
                tstamp(2);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                tstamp(3);
                matrix_slot();
                tstamp(4);
                __builtin_amdgcn_sched_barrier(0);
            } else {                           // load slot, matrix slot, counted wait, barrier
                __builtin_amdgcn_sched_barrier(0);
                tstamp(2);
                matrix_slot();
                tstamp(3);
                __builtin_amdgcn_sched_barrier(0);
                counted_wait();
                tstamp(4);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                tstamp(5);
            }
            if constexpr (STAMP && SCHED != 2 && J == 1 && KIND == 1) {
                if (stamp_on && (threadIdx.x & 63) == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) p.dbg[(1 << 19) + ((size_t)blockIdx.x * 8 + wave) * 8 + k] = ts[k];
                }
            }
        };
        // the stages of one tile, four per trip.  sa / sb: scalar offsets of this tile's stage 0; san / sbn: of the next tile's (the last two load slots request its stages 0 / 1)
        int tile_kk = 0;
        auto run_tile = [&](auto NA_c, auto NB_c, unsigned int sa, unsigned int sb, unsigned int san, unsigned int sbn) __attribute__((always_inline)) {
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            if constexpr (SCHED == 2) {
                const int ntrip = st1 >> 2;
                unsigned int ra = sa, rb = sb;
                for (int t = 0; t < ntrip; ++t) {
                    const bool last = t == ntrip - 1;   // (no in-stage stamps in this schedule: inside the one loop body they cost the loop its shape -- 3 x the cycles per stage; the tile-level stamps stay)
                    const unsigned int ha = last ? san - 256u : ra, hb = last ? sbn - 256u : rb;   // stages 2 / 3 of the last trip request stages 0 / 1 of the next tile: (san - 256) + 64 (J + 2)
                    stage(I0{}, I1{}, ra, rb, NA_c, NB_c);
                    stage(I1{}, I1{}, ra, rb, NA_c, NB_c);
                    stage(I2{}, I1{}, ha, hb, NA_c, NB_c);
                    stage(I3{}, I1{}, ha, hb, NA_c, NB_c);
                    ra += 256; rb += 256;
                }
            } else {
            using K0 = std::integral_constant<int, SCHED == 1 ? 1 : 0>;        // stage 0 of a tile: SCHED 0 requests nothing there (its stage DIST went out at the border)
            using KL0 = std::integral_constant<int, DG >= 4 ? 2 : 1>;
            using KL1 = std::integral_constant<int, DG >= 3 ? 2 : 1>;          // the kinds of the last trip's stages: stage J requests the next tile when J + DG >= 4
            using KL2 = std::integral_constant<int, DG >= 2 ? 2 : 1>;
            stage(I0{}, K0{}, sa, sb, NA_c, NB_c);
            stage(I1{}, I1{}, sa, sb, NA_c, NB_c);
            stage(I2{}, I1{}, sa, sb, NA_c, NB_c);
            stage(I3{}, I1{}, sa, sb, NA_c, NB_c);
            unsigned int ra = sa + 256, rb = sb + 256;
            for (int t = 2; t < (st1 >> 2); ++t) {
                if constexpr (STAMP) stamp_on = p.dbg && tile_kk == 1 && t == 4;
                stage(I0{}, I1{}, ra, rb, NA_c, NB_c);
                stage(I1{}, I1{}, ra, rb, NA_c, NB_c);
                stage(I2{}, I1{}, ra, rb, NA_c, NB_c);
                stage(I3{}, I1{}, ra, rb, NA_c, NB_c);
                ra += 256; rb += 256;
            }
            if constexpr (STAMP) stamp_on = false;
            stage(I0{}, KL0{}, ra, rb, NA_c, NB_c);
            if constexpr (KL1::value == 2) stage(I1{}, KL1{}, san, sbn, NA_c, NB_c); else stage(I1{}, KL1{}, ra, rb, NA_c, NB_c);
            if constexpr (KL2::value == 2) stage(I2{}, KL2{}, san, sbn, NA_c, NB_c); else stage(I2{}, KL2{}, ra, rb, NA_c, NB_c);
            stage(I3{}, I2{}, san, sbn, NA_c, NB_c);
            }
        };

        int v = blockIdx.x, tm = tm0, tn = tn0, par = 0, tile_k = 0;
        while (true) {
            stamp(tile_k, 0);
            const int vn = v + G;
            const bool has_next = vn < nblk;
            int tmn = tm, tnn = tn;          // (no next tile: the border's requests re-read this tile's first stages -- valid addresses, nobody reads them)
            if (has_next) tile_of(vn, tmn, tnn);
            const unsigned int sa = tm * a_tile, sb = tn * b_tile, san = tmn * a_tile, sbn = tnn * b_tile;
            {
                set_lane_state();
                constexpr int NA_LO = PA / 8, NA_HI = (PA + 7) / 8, NB_LO = PB / 8, NB_HI = (PB + 7) / 8;
                if (NA_HI != NA_LO && na == NA_HI) {
                    if (NB_HI != NB_LO && nb == NB_HI) run_tile(std::integral_constant<int, NA_HI>{}, std::integral_constant<int, NB_HI>{}, sa, sb, san, sbn);
                    else run_tile(std::integral_constant<int, NA_HI>{}, std::integral_constant<int, NB_LO>{}, sa, sb, san, sbn);
                } else {
                    if (NB_HI != NB_LO && nb == NB_HI) run_tile(std::integral_constant<int, NA_LO>{}, std::integral_constant<int, NB_HI>{}, sa, sb, san, sbn);
                    else run_tile(std::integral_constant<int, NA_LO>{}, std::integral_constant<int, NB_LO>{}, sa, sb, san, sbn);
                }
            }
            stamp(tile_k, 1);
            // ---- tile border.  Stage DIST of the next tile goes out before anything else (DIST 2: ring slot 2 was last read two stages ago by both groups).
            // SCHED 2: group 0, one slot ahead, takes ONE extra barrier inside its epilogue (it pairs with the barrier that ends group 1's last matrix slot), so that group 1's epilogue starts
            // beside group 0's instead of behind it; placed after the first sub-tile row's conversions and stores (border_sync(1) at the top of the loops' second trip), not at the
            // border's start: group 0 then works through group 1's last matrix slot instead of idling at the barrier for it.  Exactly one call with i == 1 (i == 2 in the two-row loop) per epilogue path.
            auto border_sync = [&](int i, int at) __attribute__((always_inline)) {
                if constexpr (SCHED == 2) { if (i == at && grp == 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } }
            };
            if constexpr (SCHED == 0) request_stage(san + 64 * DIST, sbn + 64 * DIST, DIST);   // (SCHED 1: the next tile's first load slot requests as every other one; it starts behind this border's vmcnt(0))
            const int m0 = tm * BM, n0 = tn * BN;
            // The epilogue's per-lane quantities come from an OPAQUE copy of the lane id made in every trip (see the compiler notes above).
            const int lane_e = fresh_lane();
            const int g4 = lane_e >> 4, mm = lane_e & 15;
            // this lane's 4 NTW bias values (sub-tile t: fp32 4 at 16 t + 4 g4; bf16 4 at 32 (t >> 1) + 8 g4 + 4 (t & 1) for t < 4 and at 64 + 4 g4 for the fifth)
            const char* bias_lds = smem + BIAS_LDS + par * 2048;
            float bv[4 * NTW];
#pragma unroll
            for (int e = 0; e < 4 * NTW; ++e) bv[e] = 0.f;
            if (has_bias) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const int col = wn * WW + ((p.out_f32 || t == 4) ? 16 * t + 4 * g4 : 32 * (t >> 1) + 8 * g4 + 4 * (t & 1));
                    const float4 x = *reinterpret_cast<const float4*>(bias_lds + col * 4);
                    bv[4 * t] = x.x; bv[4 * t + 1] = x.y; bv[4 * t + 2] = x.z; bv[4 * t + 3] = x.w;
                }
            }
            // ---- epilogue of tile (m0, n0), straight from the accumulators (the direct branches of gemm_ring8_kernel: same arithmetic, same stores)
            {
                const bool odd = mm & 1;
                const int row0 = m0 + row_w + mm;
                const int rowp = row0 & ~1;
                auto gst = [&](void* ptr, u32x4 vv) __attribute__((always_inline)) { *reinterpret_cast<u32x4*>(ptr) = vv; };
                if constexpr (EMODE == 1) {
                    // wqkv + RoPE + KV-cache append (modeling_internlm2.py:359-388, 233-247): see gemm_ring8_kernel
                    const int slot = (n0 >> 7) + (wn >> 1), d = 32 * (wn & 1) + 8 * g4;
                    const int gs = p.rope_G + 2, kv = slot / gs, g = slot - kv * gs;
                    const bool live = slot < p.rope_KVH * gs;
                    const bool rotate = g != gs - 1;
                    float bl[8], bh[8];
    #pragma unroll
                    for (int e = 0; e < 8; ++e) { bl[e] = (p.bias && live) ? p.bias[slot * 128 + d + e] : 0.f; bh[e] = (p.bias && live) ? p.bias[slot * 128 + d + 64 + e] : 0.f; }
                    T* const Q = reinterpret_cast<T*>(p.rope_q);
                    T* const KVc = reinterpret_cast<T*>(g == gs - 2 ? p.rope_k : p.rope_v);
                    int ps[MIW];
    #pragma unroll
                    for (int i = 0; i < MIW; ++i) {
                        const int gm = row0 + 16 * i;
                        ps[i] = (live && rotate && gm < p.M) ? min(max(p.rope_pos[gm], 0), p.rope_rows - 1) : 0;
                    }
                    float4 cn[2], sn[2];
                    auto load_cs = [&](int i, float4 (&c)[2], float4 (&s_)[2]) __attribute__((always_inline)) {
                        const float* cp = p.rope_cos + (size_t)ps[i] * 128 + d;
                        const float* sp = p.rope_sin + (size_t)ps[i] * 128 + d;
                        c[0] = *reinterpret_cast<const float4*>(cp); c[1] = *reinterpret_cast<const float4*>(cp + 4);
                        s_[0] = *reinterpret_cast<const float4*>(sp); s_[1] = *reinterpret_cast<const float4*>(sp + 4);
                    };
                    if (rotate) load_cs(0, cn, sn);
    #pragma unroll
                    for (int i = 0; i < MIW; ++i) {
                        border_sync(i, 1);
                        const float4 c0 = cn[0], c1 = cn[1], s0 = sn[0], s1 = sn[1];
                        if (rotate && i + 1 < MIW) load_cs(i + 1, cn, sn);
                        const int gm = row0 + 16 * i;
                        if (gm >= p.M || !live) continue;
                        float x[8], y[8], lo[8], hi[8];
    #pragma unroll
                        for (int e = 0; e < 8; ++e) { x[e] = acc[i][e >> 2][e & 3] + bl[e]; y[e] = acc[i][2 + (e >> 2)][e & 3] + bh[e]; }
                        if (rotate) {
                            const float cv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    #pragma unroll
                            for (int e = 0; e < 8; ++e) { lo[e] = fmaf(x[e], cv[e], -(y[e] * sv[e])); hi[e] = fmaf(y[e], cv[e], x[e] * sv[e]); }   // one rounded product + one fma, spelled out: left to -ffp-contract the one-tile and the persistent instantiations contracted differently (1 bf16 ulp in 3 of 10^6 outputs)
                        } else {
    #pragma unroll
                            for (int e = 0; e < 8; ++e) { lo[e] = x[e]; hi[e] = y[e]; }
                        }
                        T* dst;
                        if (g < p.rope_G) dst = Q + (size_t)gm * ((size_t)p.rope_KVH * p.rope_G * 128) + (size_t)(kv * p.rope_G + g) * 128 + d;
                        else {
                            const int bi = gm / p.rope_S, sq = gm - bi * p.rope_S;
                            dst = KVc + (((size_t)bi * p.rope_KVH + kv) * p.rope_cap + p.rope_pos0 + sq) * 128 + d;
                        }
                        gst(dst, (u32x4){pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(lo[4], lo[5]), pack_bf16x2(lo[6], lo[7])});
                        gst(dst + 64, (u32x4){pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]), pack_bf16x2(hi[4], hi[5]), pack_bf16x2(hi[6], hi[7])});
                    }
                } else if (!p.out_f32 && p.act != 3) {
                    // bf16 (+bias, +GELU / ReLU)
                    const int colb = n0 + wn * WW + (odd ? 32 : 0) + 8 * g4;
                    T* cp = reinterpret_cast<T*>(p.C) + (size_t)rowp * p.ldc + colb;
                    const int col5 = n0 + wn * WW + 64 + 4 * g4;
                    T* cp5 = reinterpret_cast<T*>(p.C) + (size_t)row0 * p.ldc + col5;
                    auto drain = [&](auto ACT) __attribute__((always_inline)) {
                        uint2 w5 = make_uint2(0u, 0u);
                        auto actf = [&](float x) __attribute__((always_inline)) {
                            if constexpr (decltype(ACT)::value == 1) return gelu_erfc5(x);
                            else if constexpr (decltype(ACT)::value == 2) return fmaxf(x, 0.f);
                            else return x;
                        };
    #pragma unroll
                        for (int i = 0; i < MIW; ++i) {
                            border_sync(i, 1);
                            unsigned int o[8];
    #pragma unroll
                            for (int h = 0; h < 2; ++h)
    #pragma unroll
                                for (int q = 0; q < 4; ++q)
                                    o[4 * h + q] = pack_bf16x2(actf(acc[i][2 * h + (q >> 1)][2 * (q & 1)] + bv[8 * h + 2 * q]),
                                                               actf(acc[i][2 * h + (q >> 1)][2 * (q & 1) + 1] + bv[8 * h + 2 * q + 1]));
                            u32x4 s0, s1;
                            pair_swap((u32x4){o[0], o[1], o[2], o[3]}, (u32x4){o[4], o[5], o[6], o[7]}, odd, s0, s1);
                            if (rowp + 16 * i < p.M) {
                                gst(cp + (size_t)(16 * i) * p.ldc, s0);
                                gst(cp + (size_t)(16 * i + 1) * p.ldc, s1);
                            }
                            if constexpr (NTW == 5) {
                                uint2 w;
                                w.x = pack_bf16x2(actf(acc[i][4][0] + bv[16]), actf(acc[i][4][1] + bv[17]));
                                w.y = pack_bf16x2(actf(acc[i][4][2] + bv[18]), actf(acc[i][4][3] + bv[19]));
                                if constexpr (MIW % 2 == 0) {
                                    if (i & 1) {
                                        const auto rx = __builtin_amdgcn_permlane16_swap(w5.x, w.x, false, false);
                                        const auto ry = __builtin_amdgcn_permlane16_swap(w5.y, w.y, false, false);
                                        const int r5 = row0 + 16 * (i - 1 + (g4 & 1));
                                        if (r5 < p.M)
                                            gst(reinterpret_cast<T*>(p.C) + (size_t)r5 * p.ldc + (n0 + wn * WW + 64 + 8 * (g4 >> 1)), (u32x4){rx[0], ry[0], rx[1], ry[1]});
                                    } else {
                                        w5 = w;
                                    }
                                } else {
                                    if (row0 + 16 * i < p.M) *reinterpret_cast<uint2*>(cp5 + (size_t)(16 * i) * p.ldc) = w;
                                }
                            }
                        }
                    };
                    if (p.act == 1) drain(std::integral_constant<int, 1>{});
                    else if (p.act == 2) drain(std::integral_constant<int, 2>{});
                    else drain(std::integral_constant<int, 0>{});
                } else if (NTW == 4 && !p.out_f32 && p.act == 3) {
                    // SwiGLU (modeling_internlm2.py:261-264)
                    T* cp = reinterpret_cast<T*>(p.C) + (size_t)row0 * p.ldc + (n0 >> 1) + (wn >> 1) * 64 + (wn & 1) * 32 + 8 * g4;
    #pragma unroll
                    for (int i = 0; i < MIW; ++i) {
                        border_sync(i, 1);
                        unsigned int o[4];
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float g0 = acc[i][q >> 1][2 * (q & 1)], g1 = acc[i][q >> 1][2 * (q & 1) + 1];
                            const float u0 = acc[i][2 + (q >> 1)][2 * (q & 1)], u1 = acc[i][2 + (q >> 1)][2 * (q & 1) + 1];
                            o[q] = pack_bf16x2(g0 * __builtin_amdgcn_rcpf(1.0f + __expf(-g0)) * u0, g1 * __builtin_amdgcn_rcpf(1.0f + __expf(-g1)) * u1);
                        }
                        if (row0 + 16 * i < p.M) gst(cp + (size_t)(16 * i) * p.ldc, (u32x4){o[0], o[1], o[2], o[3]});
                    }
                } else {
                    // fp32 residual stream: C = acc + bias + residual[row (mod res_row_mod)]
                    const int colp = n0 + wn * WW + (odd ? 16 : 0) + 4 * g4;
                    float* cp = reinterpret_cast<float*>(p.C) + (size_t)rowp * p.ldc + colp;
                    const int col5 = n0 + wn * WW + 64 + 4 * g4;
                    float* cp5 = reinterpret_cast<float*>(p.C) + (size_t)row0 * p.ldc + col5;
                    auto load_res = [&](int i, float4 (&r)[5]) __attribute__((always_inline)) {
    #pragma unroll
                        for (int rsel = 0; rsel < 2; ++rsel) {
                            const int gm = min(rowp + 16 * i + rsel, p.M - 1);
                            const int rr = p.res_row_mod > 0 ? gm % p.res_row_mod : gm;
                            const float* rp = p.residual + (size_t)rr * p.ldr + colp;
    #pragma unroll
                            for (int jp = 0; jp < 2; ++jp) r[2 * jp + rsel] = *reinterpret_cast<const float4*>(rp + 32 * jp);
                        }
                        if constexpr (NTW == 5) {
                            const int gm = min(row0 + 16 * i, p.M - 1);
                            const int rr = p.res_row_mod > 0 ? gm % p.res_row_mod : gm;
                            r[4] = *reinterpret_cast<const float4*>(p.residual + (size_t)rr * p.ldr + col5);
                        }
                    };
                    auto put = [&](int i, const float4 (&r)[5]) __attribute__((always_inline)) {
    #pragma unroll
                        for (int jp = 0; jp < 2; ++jp) {
                            u32x4 lo, hi, s0, s1;
    #pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                lo[e] = __float_as_uint(acc[i][2 * jp][e] + bv[8 * jp + e]);
                                hi[e] = __float_as_uint(acc[i][2 * jp + 1][e] + bv[8 * jp + 4 + e]);
                            }
                            pair_swap(lo, hi, odd, s0, s1);
                            const float4 r0 = r[2 * jp], r1 = r[2 * jp + 1];
                            s0 = (u32x4){__float_as_uint(__uint_as_float(s0[0]) + r0.x), __float_as_uint(__uint_as_float(s0[1]) + r0.y),
                                         __float_as_uint(__uint_as_float(s0[2]) + r0.z), __float_as_uint(__uint_as_float(s0[3]) + r0.w)};
                            s1 = (u32x4){__float_as_uint(__uint_as_float(s1[0]) + r1.x), __float_as_uint(__uint_as_float(s1[1]) + r1.y),
                                         __float_as_uint(__uint_as_float(s1[2]) + r1.z), __float_as_uint(__uint_as_float(s1[3]) + r1.w)};
                            if (rowp + 16 * i < p.M) {
                                *reinterpret_cast<u32x4*>(cp + (size_t)(16 * i) * p.ldc + 32 * jp) = s0;
                                *reinterpret_cast<u32x4*>(cp + (size_t)(16 * i + 1) * p.ldc + 32 * jp) = s1;
                            }
                        }
                        if constexpr (NTW == 5) {
                            if (row0 + 16 * i < p.M)
                                *reinterpret_cast<float4*>(cp5 + (size_t)(16 * i) * p.ldc) = make_float4(acc[i][4][0] + bv[16] + r[4].x, acc[i][4][1] + bv[17] + r[4].y,
                                                                                                         acc[i][4][2] + bv[18] + r[4].z, acc[i][4][3] + bv[19] + r[4].w);
                        }
                    };
                    auto drain = [&](auto RES) __attribute__((always_inline)) {
                        float4 ra[5], rb[5];
    #pragma unroll
                        for (int j = 0; j < 5; ++j) ra[j] = rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (decltype(RES)::value) load_res(0, ra);
    #pragma unroll
                        for (int i = 0; i < MIW; i += 2) {
                            border_sync(i, 2);
                            if constexpr (decltype(RES)::value) { if (i + 1 < MIW) load_res(i + 1, rb); __builtin_amdgcn_sched_barrier(0); }
                            put(i, ra);
                            if constexpr (decltype(RES)::value) { if (i + 2 < MIW) load_res(i + 2, ra); __builtin_amdgcn_sched_barrier(0); }
                            if (i + 1 < MIW) put(i + 1, rb);
                        }
                    };
                    if (p.residual) drain(std::true_type{}); else drain(std::false_type{});
                }
            }
            stamp(tile_k, 2);
            if (has_next) bias_dma(tnn * BN, par ^ 1);
            // stages 1' and 2' landed, this tile's stores acknowledged: nothing of this tile sits in front of the next one's counted waits (and no
            // LDS-DMA is in flight when the last tile's waves end)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);   // the same wait where hipcc's wait-count pass can see it: otherwise it protects the epilogue's loads / stores with waits of its own INSIDE the K loop
            stamp(tile_k, 3);
            if constexpr (SCHED == 2) { if (grp == 1) __builtin_amdgcn_s_barrier(); }   // ... and group 1 falls back one slot behind group 0 (pairs with the barrier that ends group 0's next load slot, or the kernel's last one)
            if (!has_next) break;
#pragma unroll
            for (int i = 0; i < MIW; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            v = vn; tm = tmn; tn = tnn;
            par ^= 1; ++tile_k; tile_kk = tile_k;
        }
    };
    if constexpr ((SCHED == 0 || SCHED == 2) && MI0 == MI1) core(std::integral_constant<int, MI0>{}, std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{});
    else {
        if (wm == 0) core(std::integral_constant<int, MI0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        else core(std::integral_constant<int, MI1>{}, std::integral_constant<int, MI0 * 16>{}, std::integral_constant<int, 1>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
}

template <int MI0, int MI1, int NTW, int EMODE = 0, int DIST = 2, int SCHED = 0, int ABL = 0>
static int launch_gemm_ring8p(GemmArgs a, hipStream_t stream, int max_wgs) {
    constexpr int BM = 16 * (MI0 + MI1), BN = 64 * NTW;
    constexpr int LDS = 4 * (BM + BN) * 64 + 4096;   // the ring + two bias rows
    static_assert(LDS <= 163840, "160 KiB of LDS per CU");
    constexpr bool HAS_STAMP = DIST == 2 && ABL == 0;
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring8p_kernel<MI0, MI1, NTW, EMODE, false, DIST, SCHED, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if constexpr (HAS_STAMP) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring8p_kernel<MI0, MI1, NTW, EMODE, true, DIST, SCHED, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = a.N / BN;
    a.full_tiles = a.tiles_m * a.tiles_n;
    a.ksplit = 1;
    const int grid = a.full_tiles < max_wgs ? a.full_tiles : max_wgs;
    if constexpr (HAS_STAMP) {
        if (a.dbg) {
            gemm_ring8p_kernel<MI0, MI1, NTW, EMODE, true, DIST, SCHED, ABL><<<dim3(grid), dim3(512), LDS, stream>>>(a);
            ULLSAM_LAUNCH_CHECK();
            return 0;
        }
    }
    gemm_ring8p_kernel<MI0, MI1, NTW, EMODE, false, DIST, SCHED, ABL><<<dim3(grid), dim3(512), LDS, stream>>>(a);
    ULLSAM_LAUNCH_CHECK();
    return 0;
}
