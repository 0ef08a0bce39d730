// policy_step_ws.hpp — the WAVE-SPECIALISED form of policy_step_kernel (round 5; included by policy_step.hip, which holds the
// types, the split helpers and the phase code this file re-schedules).  Same arithmetic, same results, other schedule:
//
//   one PERSISTENT workgroup per CU of 2 * NW wavefronts (NW = H / 32) that walks its tiles through a two-stage pipeline over
//   TWO A tiles in LDS:
//     helper waves  (NW of them): front phases of tile j  — masks, window descriptors, sparse encoder gather, comm, C product,
//                                 inp -> LDS (F) — then heads / draws / env.step / obs patches of tile j - 1 (B);
//     matrix waves  (NW of them): gate loop (bf16 x 9 split products) + LSTM cell epilogue of tile j (G), the obs zero fill of
//                                 the tile issued from inside that instruction stream as in policy_step_kernel.
//   A SIMD holds one matrix wave and one helper wave: its matrix pipe has exactly one owner and the phases around the MFMA
//   loops always have a wave of their own (policy_step_kernel: two workgroups per CU whose phases meet at random).
//
// gfx950 has ONE hardware barrier per workgroup and the two roles run different programs, so behind the prologue there is
// no s_barrier at all: the waves of a role meet at counter barriers in LDS, and the roles hand tiles to each other through
// monotonic LDS counters (release / acquire at workgroup scope; a spinning wave sleeps):
//     f_done   = tiles whose A tile is complete                (helpers -> matrix waves)
//     g_arrive = matrix waves that have left h' of a tile in LDS  (matrix -> helpers: NW per tile)
//     st_arrive = matrix waves whose zero stores of a tile have completed (matrix -> helpers, in front of the obs patches)
//
// Selected by ic3_policy_step when IC3_PS_WS=1 (A / B against the default kernel): recurrent policy, split gate product,
// hid 64 / 128, one communication pass, rows rewritten every step.  DESIGN.md section 10 has what it measured.
#pragma once

struct WsSync {
    int f_done, g_arrive, st_arrive, hbar, mbar, pad0, pad1, pad2;
};

// ---- role-level synchronisation through LDS ---------------------------------------------------------------------------------
__device__ __forceinline__ int ws_load(int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ws_add(int* p, int v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ws_wait(int* p, int target)
{
    while (ws_load(p) < target) __builtin_amdgcn_s_sleep(1);
}
// one lane reports for its wave: behind a wave barrier (the lanes of a wave execute together; spelled out for the scheduler)
__device__ __forceinline__ void ws_signal(int* p, int lane)
{
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) ws_add(p, 1);
}
// a barrier over the `nwaves` waves of one role: every wave adds 1 (one lane), everybody waits for epoch * nwaves
__device__ __forceinline__ void ws_role_barrier(int* ctr, int& epoch, int nwaves, int lane)
{
    epoch += nwaves;
    ws_signal(ctr, lane);
    ws_wait(ctr, epoch);
}

#ifdef IC3_PS_TRACE
#define IC3_WTR(tile, k)                                                                                                    \
    do {                                                                                                                    \
        if (a.trace && lane == 0 && wr == 0) a.trace[(size_t)(tile) * 20 + (k)] = __builtin_amdgcn_s_memrealtime();         \
    } while (0)
#else
#define IC3_WTR(tile, k) do { } while (0)
#endif

// tile list of workgroup b of G: the SMALL tile first (ids >= n_full: the envs left over behind the last balanced round — its
// front phases are the shortest, the pipeline fills sooner), then the full tiles b, b + G, ...
__device__ __forceinline__ int ws_ntiles(const StepArgs& a, int b, int G)
{
    const int nsmall = a.ntiles - a.n_full;
    return (b < a.n_full ? (a.n_full - b + G - 1) / G : 0) + (b < nsmall ? (nsmall - b + G - 1) / G : 0);
}
__device__ __forceinline__ int ws_tile(const StepArgs& a, int b, int G, int j)
{
    const int nsmall = a.ntiles - a.n_full;
    const int mine_small = b < nsmall ? (nsmall - b + G - 1) / G : 0;
    return j < mine_small ? a.n_full + b + j * G : b + (j - mine_small) * G;
}

template <int H, int KIND>
__global__ __launch_bounds__(4 * H, 1) void policy_step_ws_kernel(const StepArgs a_in)
{
    constexpr int K = 2 * H, LDA = K + 4, LDA4 = LDA / 4, BM = 64, NT = 2 * H, NW = H / 32, H4 = H / 4;
    constexpr int PER = H / 16, KB16 = K / 16;
    constexpr int SMALLW = 6 * 64 + 4;                           // sm, sscale, sact, rmask [64 each], sfm [4], sep, sts [64 each]
    IC3_DYNAMIC_LDS(float, smem);
    // The kernel arguments are read from the kernarg segment again wherever a phase needs them (reload_args: scalar loads
    // behind an opaque pointer), so that none of the ~70 dwords stays in registers across a gate loop or a tile.
    const StepArgs& a0 = a_in;
    auto tile_a = [&](int j) { return smem + (j & 1) * (BM * LDA); };   // the two A tiles
    float* const small0 = smem + 2 * BM * LDA;
    float* const shb = small0 + 2 * SMALLW;                      // [16] head / value biases
    float* const slb = shb + 16;                                 // [4H] b_ih + b_hh
    float* const shw = slb + 4 * H;                              // [16][H + 4] head / value weights (rows >= OT: zeros)
    int32_t* const tiles0 = reinterpret_cast<int32_t*>(shw + 16 * (H + 4));   // 2 x tile_words env descriptors
    WsSync* const sy = reinterpret_cast<WsSync*>(tiles0 + 2 * a0.tile_words);
    const int G = gridDim.x, b = blockIdx.x;
    const int ntl = ws_ntiles(a0, b, G);

    // ---- prologue (the only workgroup-wide barrier): counters, biases, head weights -------------------------------------------
    {
        const int t = threadIdx.x;
        if (t < 8) reinterpret_cast<int*>(sy)[t] = 0;
        if (t < 16) shb[t] = t < a0.OT ? a0.head_b[t] : 0.0f;
        for (int i = t; i < 4 * H; i += 4 * H) slb[i] = a0.l_bias[i];
        for (int i = t; i < 16 * (H + 4); i += 4 * H) {
            const int r = i / (H + 4), c = i - r * (H + 4);
            shw[i] = (r < a0.OT && c < H) ? a0.head_w[r * H + c] : 0.0f;
        }
        __syncthreads();
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#define IC3_WS_DERIVED(a)                                                                                                   \
    const int N = (a).N;                                                                                                    \
    const int WW = (KIND == IC3_ENV_PP) ? (2 * (a).pp.v + 1) * (2 * (a).pp.v + 1) : (2 * (a).tj.v + 1) * (2 * (a).tj.v + 1); \
    const int total = (a).pp.Np + (a).pp.nprey;                                                                             \
    const int nsegE = N * WW;                                                                                               \
    const int tjw = tj_tile_words(N, WW);                                                                                   \
    const float invN = 1.0f / (float)N;                                                                                     \
    const bool autor = (a).auto_reset != 0;                                                                                 \
    (void)total; (void)nsegE; (void)tjw; (void)invN; (void)autor

    if (wave < NW) {
        // =================================================================================================================
        // MATRIX WAVES: gate loop + LSTM cell epilogue of tile j, obs zero fill of tile j
        // =================================================================================================================
        const int w = wave, wr = wave, li = lane & 31, lh = lane >> 5;
        const int col = 32 * w + li;
        (void)wr;
        int mepoch = 0;
        __builtin_amdgcn_s_setprio(0);
        constexpr int GSTRIDE = NW * 64 * 16;
        ps_f32x4 zv = { 0.f, 0.f, 0.f, 0.f };
        IC3_OPAQUE_VGPR(zv);
        int st_pending = 0;                                       // a tile whose zero stores have not been reported complete yet
#pragma unroll 1
        for (int j = 0; j < ntl; ++j) {
            StepArgs a;
            reload_args(a);
            const __amdgpu_buffer_rsrc_t rg3 = make_rsrc(a.l_wp3, (uint32_t)((size_t)3 * K * 4 * H * 2));
            const int g3lane = (w * 64 + lane) * 16;
            auto wq3 = [&](int pl, int kb, int gt) __attribute__((always_inline)) {
                return __builtin_amdgcn_raw_buffer_load_b128(rg3, g3lane, ((pl * KB16 + kb) * 4 + gt) * GSTRIDE, IC3_PS_WLOAD_AUX);
            };
            const int tile_id = ws_tile(a, b, G, j);
            const TileGeom g = tile_geom<KIND>(a, tile_id);
            const bool two = g.two;
            const int rows = g.rows;
            const size_t r0 = g.r0;
            float* const As = tile_a(j);
            ps_f32x4* const As4 = reinterpret_cast<ps_f32x4*>(As);
            uint32_t* const sfm = reinterpret_cast<uint32_t*>(small0 + (j & 1) * SMALLW + 4 * 64);
            // the tile's obs slice: 1 KiB chunks dealt round-robin to the matrix waves (see policy_step_kernel)
            __amdgpu_buffer_rsrc_t zr;
            int zlane, zso;
            {
                const int first = (64 * (g.c_lo + w) - g.mis) * 16;
                zr = make_rsrc(a.obs + g.ob0 + g.ohead, (uint32_t)g.zend);
                zlane = lane * 16;
                zso = __builtin_amdgcn_readfirstlane(first);
            }
            auto zero_store = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ps_u32x4, zv), zr, zlane, zso, IC3_PS_ZSTORE_AUX);
                zso += NW * 1024;
            };
            ps_f32x16 acc[2][4];
            float cold[2][16];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[rt][gt][i] = 0.0f;
            ps_u32x4 bq[3][4];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) bq[pl][gt] = wq3(pl, 0, gt);   // (requested before the wait for the A tile)
            ws_wait(&sy->f_done, j + 1);                          // the helpers have left [inp | h] of tile j in LDS
            IC3_WTR(tile_id, 9);
            const __amdgpu_buffer_rsrc_t rc_old = make_rsrc(a.c + r0 * H, (uint32_t)rows * H * 4u);
            const int voff_old = (4 * lh * H + col) * 4;
            auto block3 = [&](auto two_c, auto s_c, auto refill_c, auto loadc_c, int kb) __attribute__((always_inline)) {
                constexpr bool TWO = decltype(two_c)::value;
                constexpr int S = decltype(s_c)::value;
                constexpr bool REFILL = decltype(refill_c)::value;
                constexpr bool LOADC = decltype(loadc_c)::value;
                constexpr int NRT = TWO ? 2 : 1;
                ps_u32x4 ap[2][3];
                ps_f32x2 xr[2][4];
                {
                    const ps_f32x4* s0 = As4 + li * LDA4 + 4 * kb + 2 * lh;
                    const ps_f32x4 x0 = s0[0], x1 = s0[1];
                    xr[0][0] = ps_f32x2{ x0[0], x0[1] }, xr[0][1] = ps_f32x2{ x0[2], x0[3] };
                    xr[0][2] = ps_f32x2{ x1[0], x1[1] }, xr[0][3] = ps_f32x2{ x1[2], x1[3] };
                    if constexpr (TWO) {
                        const ps_f32x4* s1 = As4 + (32 + li) * LDA4 + 4 * kb + 2 * lh;
                        const ps_f32x4 y0 = s1[0], y1 = s1[1];
                        xr[1][0] = ps_f32x2{ y0[0], y0[1] }, xr[1][1] = ps_f32x2{ y0[2], y0[3] };
                        xr[1][2] = ps_f32x2{ y1[0], y1[1] }, xr[1][3] = ps_f32x2{ y1[2], y1[3] };
                    }
#pragma unroll
                    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                        for (int q = 0; q < 4; ++q) ap[rt][0][q] = ps_hi_pair(xr[rt][q]);
                }
                auto products = [&](int pa, int pb, int gt) __attribute__((always_inline)) {
                    acc[0][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(ps_bf16x8, ap[0][pa]), __builtin_bit_cast(ps_bf16x8, bq[pb][gt]), acc[0][gt], 0, 0, 0);
                    if constexpr (TWO)
                        acc[1][gt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(ps_bf16x8, ap[1][pa]), __builtin_bit_cast(ps_bf16x8, bq[pb][gt]), acc[1][gt], 0, 0, 0);
                };
                auto slot = [&](int i) __attribute__((always_inline)) {
                    if (ps_zslot36(S, i)) {
                        __builtin_amdgcn_sched_barrier(0);
                        zero_store();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
#pragma unroll
                for (int pa = 0; pa < 3; ++pa) {
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt) {
                        products(pa, 0, gt);
                        slot(pa * 4 + gt);
                        if (pa < 2) {
#pragma unroll
                            for (int jj = 0; jj < NRT; ++jj) {
                                const int rt = (gt * NRT + jj) >> 2, q = (gt * NRT + jj) & 3;
                                ap[rt][pa + 1][q] = ps_next_pair(xr[rt][q], ap[rt][pa][q]);
                            }
                        } else if constexpr (REFILL) {
                            bq[0][gt] = wq3(0, kb + 1, gt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int pb = 0; pb < 3; ++pb) {
                    if (pb > 0) {
#pragma unroll
                        for (int gt = 0; gt < 4; ++gt) {
#pragma unroll
                            for (int pa = 2; pa >= 0; --pa) {
                                products(pa, pb, gt);
                                slot((pb * 4 + gt) * 3 + (2 - pa));
                            }
                            if constexpr (REFILL) bq[pb][gt] = wq3(pb, kb + 1, gt);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if constexpr (LOADC) {
#pragma unroll
                        for (int q = 0; q < 11; ++q) {
                            const int idx = 11 * pb + q;
                            if (idx < 32) {
                                const int rt = idx >> 4, reg = idx & 15;
                                if (TWO || rt == 0)
                                    cold[rt][reg] = buf_load_b32(rc_old, voff_old, (32 * rt + (reg & 3) + 8 * (reg >> 2)) * H * 4);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto gate_loop3 = [&](auto two_c, auto s_c) __attribute__((always_inline)) {
#pragma unroll 1
                for (int kb = 0; kb < KB16 - 1; ++kb) {
                    if (kb == 1 && st_pending) {
                        // behind block 0's weight waits: every store of the PREVIOUS tile has completed (one in-order counter) —
                        // at most the 12 refills of block 0 and its zero stores are younger; report it to the helpers
                        IC3_WAIT_VMEM_N(12);
                        ws_signal(&sy->st_arrive, lane);
                        st_pending = 0;
                    }
                    block3(two_c, s_c, std::true_type{}, std::false_type{}, kb);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto gate_loop3_s = [&](auto two_c) __attribute__((always_inline)) {
                // (store slots per 16-k block: 0, 8, 16 or 32 — fewer variants than policy_step_kernel: this kernel is two programs)
                const int zs = a.zs;
                if (zs >= 16) gate_loop3(two_c, std::integral_constant<int, 32>{});
                else if (zs >= 8) gate_loop3(two_c, std::integral_constant<int, 16>{});
                else if (zs >= 4) gate_loop3(two_c, std::integral_constant<int, 8>{});
                else gate_loop3(two_c, std::integral_constant<int, 0>{});
                block3(two_c, std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{}, KB16 - 1);
                __builtin_amdgcn_sched_barrier(0);
            };
            if (two) gate_loop3_s(std::true_type{});
            else gate_loop3_s(std::false_type{});
            IC3_WTR(tile_id, 10);

            // ---- LSTM cell epilogue (gate order i, f, g, o); c', h' to HBM, h' into the h half for the helpers' heads ------------
            {
                StepArgs a;                                       // (again from the kernarg segment: nothing of it crossed the loop)
                reload_args(a);
                int tidx = tile_id;
                IC3_OPAQUE_SGPR(tidx);
                const TileGeom g = tile_geom<KIND>(a, tidx);
                const bool two = g.two;
                const int rows = g.rows;
                const size_t r0 = g.r0;
                const bool autor = a.auto_reset != 0;
                // (the lane's coordinates again behind an opaque copy: the 64 LDS / buffer offsets of the element loop are
                //  loop-invariant over the tiles, and hoisted out of the tile loop they would sit in registers across every
                //  gate loop — the all-VGPR form of the accumulators does not have them to spare)
                int lane_e = lane;
                IC3_OPAQUE_VGPR(lane_e);
                const int li = lane_e & 31, lh = lane_e >> 5, col = 32 * w + li;
                const int tid = w * 64 + lane_e;
                (void)li;
                const float bi = slb[col], bf = slb[H + col], bg = slb[2 * H + col], bo = slb[3 * H + col];
                const uint32_t nrec = (uint32_t)rows * H * 4u;
                const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(a.c_out + r0 * H), 0, nrec, 0x00020000);
                const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(static_cast<void*>(a.h_out + r0 * H), 0, nrec, 0x00020000);
                const int voff = (4 * lh * H + col) * 4;
                if (autor && !a.keep_state) {
                    const unsigned long long fmask = (unsigned long long)__builtin_amdgcn_readfirstlane(sfm[0]) |
                                                     ((unsigned long long)__builtin_amdgcn_readfirstlane(sfm[1]) << 32);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            if ((fmask >> (32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh)) & 1) cold[rt][reg] = 0.0f;
                }
                ws_role_barrier(&sy->mbar, mepoch, NW, lane_e);    // every matrix wave is done reading the A tile
                auto cell = [&](auto ze_c) __attribute__((always_inline)) {
                    constexpr int ZE = decltype(ze_c)::value;
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        if (rt == 1 && !two) break;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                            const float gi = acc[rt][0][reg] + bi, gf = acc[rt][1][reg] + bf;
                            const float gg = acc[rt][2][reg] + bg, go = acc[rt][3][reg] + bo;
                            const float ig = fast_sigmoid(gi) * fast_tanh(gg);
                            const float c1 = __builtin_fmaf(fast_sigmoid(gf), cold[rt][reg], ig);
                            const float h1 = fast_sigmoid(go) * fast_tanh(c1);
                            if constexpr (ZE > 0) zero_store();
                            if constexpr (ZE > 1) zero_store();
                            const int lc = 32 * rt + (reg & 3) + 8 * (reg >> 2);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, c1), rc, voff + lc * H * 4, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, h1), rh, voff + lc * H * 4, 0, 0);
                            As[lr * LDA + H + col] = h1;
                        }
                    }
                };
                const int ze = a.zepi;
                if (ze <= 0) cell(std::integral_constant<int, 0>{});
                else if (ze == 1) cell(std::integral_constant<int, 1>{});
                else cell(std::integral_constant<int, 2>{});
                ws_signal(&sy->g_arrive, lane_e);                 // (release: the h' rows above are in LDS)
#pragma unroll 1
                for (int i = 0; i < a.zrest; ++i) zero_store();
                // ragged chunks at the ends of the body, and the <= 3 floats in front of / behind the 16-byte aligned body
                // (a.obs is never null here: ic3_policy_step takes this kernel only with the rows stored by the launch)
                ps_f32x4* const obody = reinterpret_cast<ps_f32x4*>(a.obs + g.ob0 + g.ohead);
                const int q0 = lane_e - g.mis, q1 = 64 * g.c_hi - g.mis + lane_e;
                if (w == 0 && g.mis && q0 >= 0 && q0 < g.onb) obody[q0] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
                if (w == 1 % NW && ((g.mis + g.onb) & 63) && (g.c_hi > 0 || !g.mis) && q1 >= 0 && q1 < g.onb)
                    obody[q1] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
                const int otail = (g.oL - g.ohead) & 3;
                if (tid < g.ohead) a.obs[g.ob0 + tid] = 0.f;
                if (tid < otail) a.obs[g.ob0 + g.ohead + 4 * (long long)g.onb + tid] = 0.f;
                st_pending = 1;
            }
            IC3_WTR(tile_id, 11);
        }
        if (st_pending) {                                         // the last tile: nothing behind it to wait under
            IC3_WAIT_VMEM();
            ws_signal(&sy->st_arrive, lane);
        }
    } else {
        // =================================================================================================================
        // HELPER WAVES: front phases of tile j, then heads / draws / env.step / obs patches of tile j - 1
        // =================================================================================================================
        const int tid = threadIdx.x - NT, w = wave - NW, wr = w, li = lane & 31, lh = lane >> 5;
        const int col = 32 * w + li;
        (void)wr;
        int hepoch = 0;
        __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (int j = 0; j <= ntl; ++j) {
            if (j < ntl) {
                StepArgs a;
                reload_args(a);
                IC3_WS_DERIVED(a);
                const float inv_WW = 1.0f / (float)max(WW, 1), inv_nsegE = 1.0f / (float)max(nsegE, 1);
                const BufRows encW = { __builtin_amdgcn_make_buffer_rsrc(const_cast<ps_f32x4*>(a.Wt), 0,
                                                                         (uint32_t)((size_t)a.obs_dim * H * sizeof(float)), 0x00020000),
                                       a.Wt != nullptr };
                const BufRows encL = { __builtin_amdgcn_make_buffer_rsrc(const_cast<ps_f32x4*>(a.loc_table), 0, 0x7fffffffu, 0x00020000),
                                       a.loc_table != nullptr };
                // ---------------------------------------------------------------------------------------------------------
                // F(j): the front phases of policy_step_kernel, into A tile j & 1
                // ---------------------------------------------------------------------------------------------------------
                const int tile_id = ws_tile(a, b, G, j);
                const TileGeom g = tile_geom<KIND>(a, tile_id);
                const int e0 = g.e0, nenv = g.nenv, rows = g.rows;
                const bool two = g.two;
                const size_t r0 = g.r0;
                float* const As = tile_a(j);
                ps_f32x4* const As4 = reinterpret_cast<ps_f32x4*>(As);
                float* const sm = small0 + (j & 1) * SMALLW;
                float* const sscale = sm + 64;
                int32_t* const sact = reinterpret_cast<int32_t*>(sscale + 64);
                uint32_t* const rmask = reinterpret_cast<uint32_t*>(sact + 64);
                uint32_t* const sfm = rmask + 64;
                int32_t* const sep = reinterpret_cast<int32_t*>(sfm + 4);
                int32_t* const sts = sep + 64;
                int32_t* const tile = tiles0 + (j & 1) * a.tile_words;
                IC3_WTR(tile_id, 0);
                // (buffer j & 1 was last used by tile j - 2: its B phase ran in this wave's previous iteration, and that waited
                //  for the matrix waves to be done with it)
                ps_f32x4 hv[8];
                {
                    const __amdgpu_buffer_rsrc_t rhh = __builtin_amdgcn_make_buffer_rsrc(
                        static_cast<void*>(a.h + r0 * H), 0, (uint32_t)rows * H * 4u, 0x00020000);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        hv[i] = __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rhh, tid * 16 + i * NT * 16, 0, 0));
                }
                const int rcl = min(tid, max(rows - 1, 0)), ecl = min(tid, max(nenv - 1, 0));
                int v_alive = 1, v_comm = 1, v_tsrow = 1, v_ep = 0, v_ts = 0;
                if (a.alive_in) v_alive = a.alive_in[r0 + rcl];
                if (a.comm_in) v_comm = a.comm_in[r0 + rcl];
                v_tsrow = a.tstep[e0 + div_small(rcl, invN)];
                v_ep = a.episode[e0 + ecl];
                v_ts = a.tstep[e0 + ecl];
                int32_t* sr = tile;
                int32_t* sc = tile + a.EPT * total;
                int2* ptab = reinterpret_cast<int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
                if constexpr (KIND == IC3_ENV_PP) {
                    for (int i = tid; i < nenv * total; i += NT) {
                        sr[i] = a.pp.loc_r[(size_t)e0 * total + i];
                        sc[i] = a.pp.loc_c[(size_t)e0 * total + i];
                    }
                } else {
                    for (int i = tid; i < nenv * N; i += NT) {
                        const int el = div_small(i, invN);
                        tj_tile_load_car(tj_tile_at(tile + el * tjw, N), a.tj, e0 + el, i - el * N);
                    }
                }
                if (tid < BM) {
                    const bool in = tid < rows, fr = autor && in && v_tsrow == 0;
                    sm[tid] = (in && !fr) ? (float)(v_alive * v_comm) : 0.f;
                    sact[tid] = (in && a.alive_in && !fr) ? v_alive : 1;
                    rmask[tid] = (WW <= 32) ? 0u : ~0u;
                    if (tid < nenv) {
                        sep[tid] = v_ep;
                        sts[tid] = v_ts;
                    }
                    const unsigned long long fb = __ballot(fr);
                    if (lane == 0) {
                        sfm[0] = (uint32_t)fb;
                        sfm[1] = (uint32_t)(fb >> 32);
                    }
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                IC3_WTR(tile_id, 1);
                for (int el = tid; el < nenv; el += NT) {
                    int n_alive = 0;
                    for (int jj = 0; jj < N; ++jj) n_alive += sact[el * N + jj];
                    sscale[el] = (a.mode_avg && n_alive > 1) ? 1.0f / (float)(n_alive - 1) : 1.0f;
                }
                unsigned long long fmask = 0;
                if (autor)
                    fmask = (unsigned long long)__builtin_amdgcn_readfirstlane(sfm[0]) |
                            ((unsigned long long)__builtin_amdgcn_readfirstlane(sfm[1]) << 32);
                // ---- S1: window descriptors
                {
                    int2* pt = reinterpret_cast<int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
                    for (int s = tid; s < nenv * nsegE; s += NT) {
                        const int el = div_small(s, inv_nsegE), q = s - el * nsegE;
                        int2 d;
                        if constexpr (KIND == IC3_ENV_PP) {
                            d = pp_tab_entry(tile + el * total, tile + a.EPT * total + el * total, q, a.pp.Np, total, a.pp.dim, a.pp.v);
                            pt[s] = d;
                        } else {
                            const TJTile t = tj_tile_at(tile + el * tjw, N);
                            d = tj_tab_entry(t, a.tj, q);
                            t.tab[q] = d;
                        }
                        if (d.y != 0 && WW <= 32) {
                            const int ag = div_small(q, inv_WW);
                            atomicOr(&rmask[el * N + ag], 1u << (q - ag * WW));
                        }
                    }
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                IC3_WTR(tile_id, 2);
                // ---- S2: encoder(obs) + C.bias as a sparse gather -> inp half;  S4: h -> h half
#pragma unroll IC3_PS_ENC_UNROLL
                for (int i = 0; i < 8; ++i) {
                    const int idx = tid + i * NT;
                    const int row = idx / H4, c4 = idx - row * H4;
                    ps_f32x4 v = { 0.f, 0.f, 0.f, 0.f };
                    if (row < rows) {
                        const int el = div_small(row, invN), aa = row - el * N;
                        if constexpr (KIND == IC3_ENV_PP) {
                            v = pp_encode_row_t(sr + el * total, sc + el * total, ptab + el * nsegE, aa, c4, H4, WW,
                                                a.pp.dim * a.pp.dim + 4, a.pp.dim, encW, a.enc_bias, encL, rmask[row]);
                        } else {
                            v = tj_encode_row_t(tj_tile_at(tile + el * tjw, N), a.tj, aa, c4, H4, encW, a.enc_bias, encL, rmask[row]);
                        }
                    }
                    As4[row * LDA4 + c4] = v;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int idx = tid + i * NT;
                    const int row = idx / H4, c4 = idx - row * H4;
                    As4[row * LDA4 + H4 + c4] = (autor && !a.keep_state && ((fmask >> row) & 1)) ? ps_f32x4{ 0.f, 0.f, 0.f, 0.f } : hv[i];
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                IC3_WTR(tile_id, 3);
                // ---- S3: the encoder output moves into the accumulators of the C product
                ps_f32x16 accC[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    if (rt == 1 && !two) break;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        accC[rt][reg] = As[lr * LDA + col];
                    }
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);    // every helper wave has its share of the encoder output
                IC3_WTR(tile_id, 4);
                if (!a.comm_zero) {
                    // ---- S5: comm_j = m_j (S_e - m_j h_j) scale_e -> inp half
                    {
                        const int c4 = tid % H4;
                        for (int el = tid / H4; el < nenv; el += NT / H4) {
                            const ps_f32x4* hp = As4 + (el * N) * LDA4 + H4 + c4;
                            const float scl = sscale[el];
                            ps_f32x4 S = { 0.f, 0.f, 0.f, 0.f };
                            for (int i = 0; i < N; ++i) S += sm[el * N + i] * hp[i * LDA4];
                            for (int jj = 0; jj < N; ++jj) {
                                const float m = sm[el * N + jj];
                                As4[(el * N + jj) * LDA4 + c4] = m * (S - m * hp[jj * LDA4]) * scl;
                            }
                        }
                        for (int idx = rows * H4 + tid; idx < BM * H4; idx += NT) {
                            const int row = idx / H4, c4p = idx - row * H4;
                            As4[row * LDA4 + c4p] = ps_f32x4{ 0.f, 0.f, 0.f, 0.f };
                        }
                    }
                    constexpr int KBC = H / 8, CH = (KBC < 8) ? KBC : 8, NCH = KBC / CH;
                    const __amdgpu_buffer_rsrc_t rcw = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<ps_f32x4*>(a.c_wp), 0, (uint32_t)((size_t)H * H * sizeof(float)), 0x00020000);
                    const int wlane = (col * 2 + lh) * 16;
                    auto cwp = [&](int k) {
                        return __builtin_bit_cast(ps_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rcw, wlane, k * (H * 2 * 16), 0));
                    };
                    ps_f32x4 cb[2][CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k) cb[0][k] = cwp(k);
                    ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                    IC3_WTR(tile_id, 6);
                    // ---- S6: accC (= enc) += comm . C.weight^T
                    auto cprod = [&](auto two_c) __attribute__((always_inline)) {
                        constexpr bool TWO = decltype(two_c)::value;
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) {
                            if (ch + 1 < NCH) {
#pragma unroll
                                for (int k = 0; k < CH; ++k) cb[(ch + 1) & 1][k] = cwp((ch + 1) * CH + k);
                            }
#pragma unroll
                            for (int k = 0; k < CH; ++k) {
                                const int kb = ch * CH + k;
                                const ps_f32x4 a0 = As4[li * LDA4 + 2 * kb + lh];
                                ps_f32x4 a1;
                                if constexpr (TWO) a1 = As4[(32 + li) * LDA4 + 2 * kb + lh];
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) {
                                    accC[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[jj], cb[ch & 1][k][jj], accC[0], 0, 0, 0);
                                    if constexpr (TWO) accC[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[jj], cb[ch & 1][k][jj], accC[1], 0, 0, 0);
                                }
                            }
                        }
                    };
                    if (two) cprod(std::true_type{});
                    else cprod(std::false_type{});
                    ws_role_barrier(&sy->hbar, hepoch, NW, lane);   // every helper wave has read the comm tile
                    IC3_WTR(tile_id, 7);
                }
                // ---- S7: inp = enc + C.bias + C(comm) -> inp half
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    if (rt == 1 && !two) break;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int lr = 32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                        As[lr * LDA + col] = accC[rt][reg];
                    }
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                if (w == 0) ws_signal(&sy->f_done, lane);         // tile j is the matrix waves'
                IC3_WTR(tile_id, 8);
            }
            if (j >= 1) {
                // ---------------------------------------------------------------------------------------------------------
                // B(j - 1): heads + value, log-softmax + draws, env.step, obs patches of tile j - 1
                // ---------------------------------------------------------------------------------------------------------
                StepArgs a;
                reload_args(a);
                IC3_WS_DERIVED(a);
                const int jb = j - 1;
                const int tile_id = ws_tile(a, b, G, jb);
                const TileGeom g = tile_geom<KIND>(a, tile_id);
                const int e0 = g.e0, nenv = g.nenv, rows = g.rows;
                const size_t r0 = g.r0;
                float* const As = tile_a(jb);
                ps_f32x4* const As4 = reinterpret_cast<ps_f32x4*>(As);
                float* const sm = small0 + (jb & 1) * SMALLW;
                int32_t* const sact = reinterpret_cast<int32_t*>(sm + 128);
                int32_t* const sep = reinterpret_cast<int32_t*>(sm + 4 * 64 + 4);
                int32_t* const sts = sep + 64;
                int32_t* const tile = tiles0 + (jb & 1) * a.tile_words;
                ws_wait(&sy->g_arrive, NW * (jb + 1));            // h' of tile j - 1 is in the h half (all matrix waves)
                IC3_WTR(tile_id, 12);
                // ---- S10: heads + value head on v_mfma_f32_16x16x4_f32; logits of row r -> rows [0, ..) of the inp half:
                //      z(r, o) = As[(r / PER) * LDA + (r % PER) * 16 + o]   (the head weights have an LDS block of their own)
                {
                    const int l16 = lane & 15, kq = lane >> 4;
                    const float hb = shb[l16];
                    const ps_f32x4* shw4 = reinterpret_cast<const ps_f32x4*>(shw);
                    for (int rtile = w; rtile < BM / 16; rtile += NW) {
                        if (16 * rtile >= rows) break;
                        ps_f32x4 z = { hb, hb, hb, hb };
                        const ps_f32x4* xa = As4 + (16 * rtile + l16) * LDA4 + H4 + kq;
                        const ps_f32x4* wb = shw4 + l16 * ((H + 4) / 4) + kq;
#pragma unroll 4
                        for (int sg = 0; sg < H / 16; ++sg) {
                            const ps_f32x4 x4 = xa[4 * sg], w4 = wb[4 * sg];
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) z = __builtin_amdgcn_mfma_f32_16x16x4f32(x4[jj], w4[jj], z, 0, 0, 0);
                        }
                        if (l16 < a.OT) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int r = 16 * rtile + 4 * kq + v;
                                As[(r / PER) * LDA + (r % PER) * 16 + l16] = z[v];
                            }
                        }
                    }
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                IC3_WTR(tile_id, 13);
                // ---- S11: log_softmax per head + the action draws
                {
                    const int sizes[4] = { a.a0, a.a1, a.a2, a.a3 };
                    const int R = a.E * N;
                    const float inv_nh1 = 1.0f / (float)(a.nheads + 1);
                    for (int task = tid; task < rows * (a.nheads + 1); task += NT) {
                        const int tr = div_small(task, inv_nh1), hd = task - tr * (a.nheads + 1);
                        const size_t grow = r0 + tr;
                        const float* z = As + (tr / PER) * LDA + (tr % PER) * 16;
                        float* orow = a.out + grow * a.OT;
                        int off = 0;
                        for (int i = 0; i < hd && i < a.nheads; ++i) off += sizes[i];
                        if (hd == a.nheads) {
                            orow[off] = z[off];
                            continue;
                        }
                        const int A = sizes[hd];
                        float mx = -INFINITY;
                        for (int o = 0; o < A; ++o) mx = fmaxf(mx, z[off + o]);
                        float sum = 0.0f;
                        for (int o = 0; o < A; ++o) sum += __builtin_amdgcn_exp2f(1.4426950408889634f * (z[off + o] - mx));
                        const float lse = mx + 0.6931471805599453f * __builtin_amdgcn_logf(sum);
                        for (int o = 0; o < A; ++o) orow[off + o] = z[off + o] - lse;
                        const int el = div_small(tr, invN), n = tr - el * N;
                        const int e = e0 + el;
                        const uint32_t x = philox_x24(a.seed, a.gid0 + (uint32_t)e, DOMAIN_SAMPLE, (uint32_t)sep[el],
                                                      (uint32_t)sts[el], (uint32_t)(hd * N + n));
                        const float u = (float)x * (1.0f / 16777216.0f);
                        float cdf = 0.0f;
                        int act = A - 1;
                        for (int o = 0; o < A - 1; ++o) {
                            cdf += expf(z[off + o] - lse);
                            if (u < cdf) {
                                act = o;
                                break;
                            }
                        }
                        a.action[(size_t)hd * R + grow] = act;
                        if (hd == 0) sact[tr] = act;
                    }
                }
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                IC3_WTR(tile_id, 14);
                // ---- S12: env.step for the tile's envs
                {
                    const int lgG = __builtin_ctz(a.G);
                    for (int base = 0; base < a.EPT * a.G; base += NT) {
                        const int lt = base + tid;
                        const int el = lt >> lgG, n = lt - (el << lgG);
                        const int e = el < nenv ? e0 + el : a.E;
                        if constexpr (KIND == IC3_ENV_PP) {
                            pp_step_lanes(a.pp, a.so, e, n, a.E, a.G, [&]() { return sact[el * N + n]; });
                        } else {
                            tj_step_lanes(a.tj, a.so, e, n, a.E, a.G, [&]() { return sact[el * N + n]; });
                        }
                    }
                }
                IC3_WTR(tile_id, 15);
                // ---- obs patches: behind every zero store of the tile (the matrix waves report them complete)
                ws_wait(&sy->st_arrive, NW * (jb + 1));
                IC3_WTR(tile_id, 16);
                {
                    float* orow0 = a.obs + g.ob0;
                    if constexpr (KIND == IC3_ENV_PP) {
                        const int2* ptab = reinterpret_cast<const int2*>(tile + ((2 * a.EPT * total + 3) & ~3));
                        const int vocab = a.pp.dim * a.pp.dim + 4;
                        for (int sg = tid; sg < nenv * nsegE; sg += NT) pp_obs_patch(orow0 + (size_t)sg * vocab, ptab[sg], vocab);
                    } else {
                        const int obs_dim = a.obs_dim;
                        for (int sg = tid; sg < nenv * (nsegE + N); sg += NT) {
                            const int el = div_small(sg, 1.0f / (float)(nsegE + N)), q = sg - el * (nsegE + N);
                            const TJTile t = tj_tile_at(tile + el * tjw, N);
                            tj_obs_patch(t, a.tj, orow0 + (size_t)el * N * obs_dim, obs_dim, WW, q);
                        }
                    }
                }
                // (the next iteration's F writes the OTHER buffer; this buffer's next writer is F(j + 1), behind this point of
                //  every helper wave only after the role barrier of its first phase)
                ws_role_barrier(&sy->hbar, hepoch, NW, lane);
                IC3_WTR(tile_id, 17);
            }
        }
    }
}
