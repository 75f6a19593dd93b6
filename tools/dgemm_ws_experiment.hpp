// dgemm_ws_experiment.hpp — NOT part of the library: the wave-specialised persistent fp64 GEMM built and measured in round 4
// (consumers = MFMA only, helpers = all memory traffic, LDS sequence-number protocol, C tile prefetched into the helpers' registers).
// Correct (the 63 dgemm tests of tests/test_gpu_syevd.py pass with it, no wait ever expired), but slower than the ring kernel of
// csrc/dgemm.hpp on every shape: square 8192^3 37.6 TF against 60, rank-128 update 23.8 against 40, Q1's K = 256 update 30 against 48
// (profiles/r04_gemm_experiments.txt).  What it showed: with ONE 128 x 128 tile per CU the four helper waves' LDS-DMA moves a 16 KB chunk
// in ~1 700 cycles (9.6 B/clk/CU, the CU's load rate for operands that miss L2) while the products of a chunk take 2 048 — a 128 x 128 fp64
// tile needs 8 B/clk/CU of operand traffic at the full MFMA rate, 80 % of what a CU can load, so one workgroup per CU starves where two
// (the ring kernel) overlap each other's gaps.  Kept for the record; to try again it needs a 256 x 128 tile (6 B/clk).
// It was included from csrc/dgemm.hpp between the ring kernel and splitk_reduce_kernel, dispatched from dgemm_ex (PG_DGEMM_WS).
// ==== wave-specialised persistent kernel (r4) =======================================================================================
// What the in-kernel stamps and the depth sweep of the rank-2b update said about the ring kernel above (tools/ring_stamps.py,
// tools/bench_update_k.py): its time is the SUM of the C tile's memory time (0.158 ms at m = 9 984: 5.1 TB/s, the HBM rate) and of the
// products' time (0.164 ms) — no overlap: every workgroup of the chip computes, then every workgroup moves its tile; and a wave that
// waits, crosses a barrier and issues its own DMA between two chunks leaves the matrix pipe idle meanwhile (3 200 cycles per chunk of
// 2 048 cycles of products with one workgroup per CU).  Here the two jobs belong to different waves of ONE 512-thread workgroup per
// CU that stays for many tiles:
//   consumers (waves 0-3, one per SIMD, 64 x 64 of the 128 x 128 tile each): fragment reads and MFMAs, nothing else.  They learn that a
//       chunk has landed from a word in LDS and announce with another that a slot may be refilled; at the end of a tile they put the
//       accumulators (alpha applied) into 16-row staging buffers and go on with the next tile, whose first chunks are already there.
//   helpers (waves 4-7): all memory traffic.  LDS-DMA of the operand chunks four to five chunks ahead across tile boundaries; the C
//       tile (beta != 0) is read into their own REGISTERS — 128 KB per workgroup, the one buffer of that size the CU has left — while
//       the tile's products run; staged rows + beta C leave as whole 1 KB rows.  The tile's memory traffic is spread over its products.
// Protocol words are monotonically increasing sequence numbers in LDS (no resets, no ABA); every spin is bounded and raises a flag
// that makes every wave of the workgroup leave (tests read it: pgx_ws_aborts).
constexpr int WRS = 5, WSG = 4;
constexpr int WS_SLOT = 16384;
constexpr int WS_LDS = WRS * WS_SLOT + WSG * WS_SLOT + 512;
struct WsCtl {
    int full[WRS][4];       // [slot][helper]: chunk sequence number + 1 whose quarter that helper's DMA has landed in the slot
    int freed[WRS][4];      // [slot][consumer]: sequence number + 1 of the chunk that consumer has read from the slot
    int sfull[WSG][2];      // [staging buffer][wn]: per-half group sequence number + 1 written there
    int sfree[WSG][4];      // [staging buffer][helper]: per-half group sequence number + 1 stored from there
    int abort;
};
static __device__ int g_ws_aborts;       // workgroups that gave up a wait (never expected)
// (readfirstlane: every lane reads the same word — the value is wave-uniform, and the compiler has to know it, or the helpers' whole state
// machine becomes per-lane control flow with exec masks and spills)
#define WS_LD(p) __builtin_amdgcn_readfirstlane(__hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#define WS_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
constexpr int WS_SPIN_MAX = 1 << 24;
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));
// the smallest of the four (two) words of a protocol entry, read with ONE LDS access (four dependent ds_read + readfirstlane round
// trips per check were most of the helpers' 5 000 cycles per chunk)
__device__ __forceinline__ int ws_min4(const int *p)
{
    const intx4 v = *reinterpret_cast<const volatile intx4 *>(p);
    const int a = v.x < v.y ? v.x : v.y, b = v.z < v.w ? v.z : v.w;
    return __builtin_amdgcn_readfirstlane(a < b ? a : b);
}
__device__ __forceinline__ int ws_min2(const int *p)
{
    const intx2 v = *reinterpret_cast<const volatile intx2 *>(p);
    return __builtin_amdgcn_readfirstlane(v.x < v.y ? v.x : v.y);
}

// the two roles are separate (not inlined) functions: each gets a register allocation of its own — inlined into one kernel body the
// consumers' loop spilled fragment addresses to scratch and reloaded them behind an s_waitcnt vmcnt(0) in every chunk
template <bool AKM>
static __device__ __attribute__((noinline)) void ws_consumer(const DgemmParams &gp_in, unsigned char *wsm)
{
    // the parameter block by VALUE in registers: read through the reference, every field access was a flat load followed by
    // s_waitcnt vmcnt(0) — i.e. a wait for every DMA in flight — after each compiler barrier (2 600 cycles per chunk issue; r4)
    struct {
        long long M, N, K, lda, ldb, ldc; const double *A, *B; double *C; double alpha, beta; int lower, kxorB; long long *stamps;
    } gp;
    gp.M = gp_in.M; gp.N = gp_in.N; gp.K = gp_in.K; gp.lda = gp_in.lda; gp.ldb = gp_in.ldb; gp.ldc = gp_in.ldc; gp.A = gp_in.A; gp.B = gp_in.B; gp.C = gp_in.C;
    gp.alpha = gp_in.alpha; gp.beta = gp_in.beta; gp.lower = gp_in.lower; gp.kxorB = gp_in.kxorB; gp.stamps = gp_in.stamps;
    unsigned char *const ring = wsm, *const stage = wsm + WRS * WS_SLOT;
    WsCtl *const ctl = reinterpret_cast<WsCtl *>(wsm + (WRS + WSG) * WS_SLOT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;     // (the wave number: uniform, and said so)
    const long long M = gp.M, N = gp.N;
    const int tiles_m = (int)((M + 127) / 128), tiles_n = (int)((N + 127) / 128);
    const int T = (gp.lower == 1) ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    const int G = gridDim.x;
    const int ntile = (T - (int)blockIdx.x + G - 1) / G;            // tiles of this workgroup: blockIdx.x, + G, + 2 G ...
    const int NCH = (int)(gp.K / RBK);
    auto tile_rc = [&](int it, int &tm, int &tn) {
        const int lid = (int)blockIdx.x + it * G;
        if (gp.lower == 1) {
            int t_ = (int)((sqrt(8.0 * (double)lid + 1.0) - 1.0) * 0.5);
            while ((t_ + 1) * (t_ + 2) / 2 <= lid) t_++;
            while (t_ * (t_ + 1) / 2 > lid) t_--;
            tm = t_; tn = lid - t_ * (t_ + 1) / 2;
        } else { tm = lid / tiles_n; tn = lid % tiles_n; }
    };
    // staging buffer and per-half sequence number of row group g (0..7) of this workgroup's it-th tile: the upper half of the tile
    // (g < 4) alternates between buffers 0 and 1, the lower half between 2 and 3
    auto grp_buf = [&](int g) { return 2 * (g >> 2) + (g & 1); };
    auto grp_seq = [&](int it, int g) { return 4 * it + (g & 3); };

        // =========================================================== consumers
        const int wm = wave >> 1, wn = wave & 1;
        int q = 0;
        bool dead = false;
        const bool ws_diag = gp.stamps != nullptr && blockIdx.x == 100 && wave == 0;
        long long t_wait = 0, n_spin = 0;
        const long long t_begin = ws_diag ? __builtin_amdgcn_s_memtime() : 0;
        for (int it = 0; it < ntile && !dead; it++) {
            int tm, tn;
            tile_rc(it, tm, tn);
            const bool skip = (gp.lower == 2) && ((long long)tn * 128 >= (long long)tm * 128 + 128);   // above the diagonal of a rectangular lower C
            doublex4 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[i][j][e] = 0.0;
            for (int t = 0; t < NCH && !dead; t++, q++) {
                const int slot = q % WRS;
                const unsigned char *sl = ring + slot * WS_SLOT;
                {   // the chunk's four quarters have landed
                    int spins = 0;
                    const long long tw0 = ws_diag ? __builtin_amdgcn_s_memtime() : 0;
                    for (;;) {
                        if (ws_min4(&ctl->full[slot][0]) > q) break;
                        if (WS_LD(&ctl->abort) != 0 || ++spins > WS_SPIN_MAX) { dead = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (dead) break;
                    asm volatile("" ::: "memory");
                    if (ws_diag) { t_wait += __builtin_amdgcn_s_memtime() - tw0; n_spin += spins; }
                }
                if (skip) { if (lane == 0) WS_ST(&ctl->freed[slot][wave], q + 1); continue; }
                double a0[4], b0[4], a1[4], b1[4];
                auto frag_a = [&](int ks, int i) {
                    const int k = 4 * ks + (lane >> 4), ml = wm * 64 + 16 * i + (lane & 15);
                    const int off = AKM ? k * 1024 + ((ml * 8) ^ ((k & 1) * 128)) : ml * 64 + (((k >> 1) ^ ((ml >> 2) & 3)) * 16) + (k & 1) * 8;
                    return *reinterpret_cast<const double *>(sl + off);
                };
                auto frag_b = [&](int ks, int j) {
                    const int k = 4 * ks + (lane >> 4), nl = wn * 64 + 16 * j + (lane & 15);
                    return *reinterpret_cast<const double *>(sl + 8192 + k * 1024 + ((nl * 8) ^ ((k & 1) * 128)));
                };
#pragma unroll
                for (int i = 0; i < 4; i++) a0[i] = frag_a(0, i);
#pragma unroll
                for (int j = 0; j < 4; j++) b0[j] = frag_b(0, j);
#pragma unroll
                for (int i = 0; i < 4; i++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[i], b0[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == 0) {
#pragma unroll
                        for (int i2 = 0; i2 < 4; i2++) a1[i2] = frag_a(1, i2);
#pragma unroll
                        for (int j = 0; j < 4; j++) b1[j] = frag_b(1, j);
                    }
                    // the LDS unit serves a wave's operations in order: this word is written after the reads above have been served
                    if (i == 1) { asm volatile("" ::: "memory"); if (lane == 0) WS_ST(&ctl->freed[slot][wave], q + 1); }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[i], b1[j], acc[i][j], 0, 0, 0);
            }
            if (dead) break;
            // the tile's rows to the staging buffers: this wave's four groups of 16 rows (64 columns of each)
#pragma unroll
            for (int gi = 0; gi < 4; gi++) {
                const int g = 4 * wm + gi, b = grp_buf(g), hs = grp_seq(it, g);
                if (hs >= 2) {      // the buffer's previous group (two back in this half's sequence) has left
                    int spins = 0;
                    for (;;) {
                        if (ws_min4(&ctl->sfree[b][0]) > hs - 2) break;
                        if (WS_LD(&ctl->abort) != 0 || ++spins > WS_SPIN_MAX) { dead = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    asm volatile("" ::: "memory");
                }
                if (dead) break;
                unsigned char *sb = stage + b * WS_SLOT;
                if (!skip) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int rl = (lane >> 4) + 4 * e, col = wn * 64 + 16 * j + (lane & 15);
                            *reinterpret_cast<double *>(sb + rl * 1024 + ((col * 8) ^ ((rl & 1) * 128))) = gp.alpha * acc[gi][j][e];
                        }
                }
                asm volatile("" ::: "memory");
                if (lane == 0) WS_ST(&ctl->sfull[b][wn], hs + 1);
            }
        }
        if (ws_diag && lane == 0) { gp.stamps[40] = t_wait; gp.stamps[41] = __builtin_amdgcn_s_memtime() - t_begin; gp.stamps[46] = n_spin; gp.stamps[47] = q; }
        if (dead && lane == 0) { WS_ST(&ctl->abort, 1); if (wave == 0) atomicAdd(&g_ws_aborts, 1); }
}

template <bool AKM>
static __device__ __attribute__((noinline)) void ws_helper(const DgemmParams &gp_in, unsigned char *wsm)
{
    // the parameter block by VALUE in registers: read through the reference, every field access was a flat load followed by
    // s_waitcnt vmcnt(0) — i.e. a wait for every DMA in flight — after each compiler barrier (2 600 cycles per chunk issue; r4)
    struct {
        long long M, N, K, lda, ldb, ldc; const double *A, *B; double *C; double alpha, beta; int lower, kxorB; long long *stamps;
    } gp;
    gp.M = gp_in.M; gp.N = gp_in.N; gp.K = gp_in.K; gp.lda = gp_in.lda; gp.ldb = gp_in.ldb; gp.ldc = gp_in.ldc; gp.A = gp_in.A; gp.B = gp_in.B; gp.C = gp_in.C;
    gp.alpha = gp_in.alpha; gp.beta = gp_in.beta; gp.lower = gp_in.lower; gp.kxorB = gp_in.kxorB; gp.stamps = gp_in.stamps;
    unsigned char *const ring = wsm, *const stage = wsm + WRS * WS_SLOT;
    WsCtl *const ctl = reinterpret_cast<WsCtl *>(wsm + (WRS + WSG) * WS_SLOT);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;     // (the wave number: uniform, and said so)
    const long long M = gp.M, N = gp.N;
    const int tiles_m = (int)((M + 127) / 128), tiles_n = (int)((N + 127) / 128);
    const int T = (gp.lower == 1) ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    const int G = gridDim.x;
    const int ntile = (T - (int)blockIdx.x + G - 1) / G;            // tiles of this workgroup: blockIdx.x, + G, + 2 G ...
    const int NCH = (int)(gp.K / RBK);
    auto tile_rc = [&](int it, int &tm, int &tn) {
        const int lid = (int)blockIdx.x + it * G;
        if (gp.lower == 1) {
            int t_ = (int)((sqrt(8.0 * (double)lid + 1.0) - 1.0) * 0.5);
            while ((t_ + 1) * (t_ + 2) / 2 <= lid) t_++;
            while (t_ * (t_ + 1) / 2 > lid) t_--;
            tm = t_; tn = lid - t_ * (t_ + 1) / 2;
        } else { tm = lid / tiles_n; tn = lid % tiles_n; }
    };
    // staging buffer and per-half sequence number of row group g (0..7) of this workgroup's it-th tile: the upper half of the tile
    // (g < 4) alternates between buffers 0 and 1, the lower half between 2 and 3
    auto grp_buf = [&](int g) { return 2 * (g >> 2) + (g & 1); };
    auto grp_seq = [&](int it, int g) { return 4 * it + (g & 3); };

    const int h = wave - 4;
    const bool rdC = gp.beta != 0.0;
    const long long colA_max = (M - 1) & ~1LL, colB_max = (N - 1) & ~1LL;
    int qi = 0, qp = 0;                  // next chunk (global sequence number) to issue / to publish
    int n_ops = 0;                       // vector-memory operations this wave has issued so far
    int mark[WRS];                       // n_ops right after the DMAs of the chunk in each slot
#pragma unroll
    for (int s_ = 0; s_ < WRS; s_++) mark[s_] = 0;
    const int QT = ntile * NCH;
    int itile = -1;                      // tile the issue offsets below belong to
    long long offA[2] = {0, 0}, offB[2] = {0, 0};
    bool dead = false;
    auto set_issue_tile = [&](int it) {
        int tm, tn;
        tile_rc(it, tm, tn);
        const long long m0 = (long long)tm * 128, n0 = (long long)tn * 128;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int a = 2 * h + u;
            if (AKM) {
                long long col = m0 + 2 * (lane ^ (8 * (a & 1)));
                col = col < colA_max ? col : colA_max;
                offA[u] = (long long)a * gp.lda + col;
            } else {
                const int r = 16 * a + (lane >> 2), c = lane & 3;
                long long row = m0 + r;
                row = row < M ? row : M - 1;
                offA[u] = row * gp.lda + 2 * (c ^ ((r >> 2) & 3));
            }
            long long cb = n0 + 2 * (lane ^ (8 * (a & 1)));
            cb = cb < colB_max ? cb : colB_max;
            offB[u] = cb;
        }
        itile = it;
    };
    // keep the ring fed: issue the next chunk when its slot has been read by all four consumers; publish the oldest chunk in flight once
    // two younger ones are in flight behind it (or nothing is left to issue)
    const bool ws_diag = gp.stamps != nullptr && blockIdx.x == 100 && h == 0;
    long long t_vm = 0, n_pump = 0, n_issue = 0, t_issue = 0;
    const long long t_begin = ws_diag ? __builtin_amdgcn_s_memtime() : 0;
    int it_i = 0, t_i = 0, slot_i = 0, slot_p = 0;      // tile / chunk-in-tile / slot of the next chunk to issue; slot of the next to publish
    auto pump = [&]() {
        bool did = false;
        if (ws_diag) n_pump++;
        // (the protocol state is wave-uniform; said explicitly, so that it lives in scalar registers and branches are scalar)
        qi = __builtin_amdgcn_readfirstlane(qi); qp = __builtin_amdgcn_readfirstlane(qp); n_ops = __builtin_amdgcn_readfirstlane(n_ops);
        it_i = __builtin_amdgcn_readfirstlane(it_i); t_i = __builtin_amdgcn_readfirstlane(t_i);
        slot_i = __builtin_amdgcn_readfirstlane(slot_i); slot_p = __builtin_amdgcn_readfirstlane(slot_p);
#pragma unroll
        for (int s_ = 0; s_ < WRS; s_++) mark[s_] = __builtin_amdgcn_readfirstlane(mark[s_]);
        if (qi < QT) {
            const bool ok = qi < WRS || ws_min4(&ctl->freed[slot_i][0]) > qi - WRS;      // the slot's previous chunk has been read by all four consumers
            if (ok) {
                asm volatile("" ::: "memory");
                const long long ti0 = ws_diag ? __builtin_amdgcn_s_memtime() : 0;
                if (it_i != itile) set_issue_tile(it_i);
                const long long k0 = (long long)t_i * RBK;
                unsigned char *sl = ring + slot_i * WS_SLOT;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int a = 2 * h + u;
                    const double *srcA = AKM ? gp.A + k0 * gp.lda + offA[u] : gp.A + offA[u] + k0;
                    ring_dma16(srcA, sl + a * 1024);
                    ring_dma16(gp.B + (((k0 + a) ^ (long long)gp.kxorB)) * gp.ldb + offB[u], sl + 8192 + a * 1024);
                }
                n_ops += 4;
#pragma unroll
                for (int s_ = 0; s_ < WRS; s_++) if (s_ == slot_i) mark[s_] = n_ops;
                qi++;
                if (++t_i == NCH) { t_i = 0; it_i++; }
                if (++slot_i == WRS) slot_i = 0;
                did = true;
                if (ws_diag) { n_issue++; t_issue += __builtin_amdgcn_s_memtime() - ti0; }
            }
        }
        if (qp < qi && (qi - qp >= 3 || qi == QT)) {
            int mk = 0;
#pragma unroll
            for (int s_ = 0; s_ < WRS; s_++) if (s_ == slot_p) mk = mark[s_];
            const int younger = __builtin_amdgcn_readfirstlane(n_ops - mk);
            const long long tv0 = ws_diag ? __builtin_amdgcn_s_memtime() : 0;
            if (younger == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // the steady state: two younger chunks, nothing else
            else ring_wait_vm(younger < 30 ? younger : 30);
            if (ws_diag) t_vm += __builtin_amdgcn_s_memtime() - tv0;
            if (lane == 0) WS_ST(&ctl->full[slot_p][h], qp + 1);
            qp++;
            if (++slot_p == WRS) slot_p = 0;
            did = true;
        }
        return did;
    };
    // this helper's share of a tile's C: rows 4 h .. 4 h + 3 of each of the 8 groups, 16 bytes per lane and row
    doublex2 creg[8][4];
    auto c_ptr = [&](int tm, int tn, int g, int rr, bool &rowok, bool &ok) {
        const long long row = (long long)tm * 128 + 16 * g + 4 * h + rr, col = (long long)tn * 128 + 2 * lane;
        rowok = row < M;                     // uniform over the wave
        ok = rowok && col + 1 < N;
        return gp.C + (ok ? row * gp.ldc + col : 0);
    };
    auto load_c = [&](int it) {
        int tm, tn;
        tile_rc(it, tm, tn);
#pragma unroll
        for (int g = 0; g < 8; g++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                bool rowok, ok;
                const double *p = c_ptr(tm, tn, g, rr, rowok, ok);
                // (address space 1 said explicitly: a flat_load would return out of order with the DMAs and break the counted waits)
                creg[g][rr] = *reinterpret_cast<const __attribute__((address_space(1))) doublex2 *>(reinterpret_cast<uintptr_t>(p));   // out-of-range lanes read element 0 and are never stored
            }
        n_ops += 32;                         // issued for every lane whatever the predicate: exact
    };
    if (rdC && ntile > 0) load_c(0);
    for (int it = 0; it < ntile && !dead; it++) {
        int tm, tn;
        tile_rc(it, tm, tn);
        const bool skip = (gp.lower == 2) && ((long long)tn * 128 >= (long long)tm * 128 + 128);
#pragma unroll
        for (int gg = 0; gg < 8; gg++) {
            const int g = (gg >> 1) + 4 * (gg & 1);        // 0, 4, 1, 5, ...: the two halves of the tile in turn
            const int b = grp_buf(g), hs = grp_seq(it, g);
            int spins = 0;
            for (;;) {       // both consumers of the group's half have written it; meanwhile the ring is kept fed
                if (ws_min2(&ctl->sfull[b][0]) > hs) break;
                if (!pump()) {
                    if (WS_LD(&ctl->abort) != 0 || ++spins > WS_SPIN_MAX) { dead = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (dead) break;
            asm volatile("" ::: "memory");
            const unsigned char *sb = stage + b * WS_SLOT;
            if (!skip) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int row = 4 * h + rr;
                    doublex2 v = *reinterpret_cast<const doublex2 *>(sb + row * 1024 + ((lane * 16) ^ ((row & 1) * 128)));
                    bool rowok, ok;
                    const double *p = c_ptr(tm, tn, g, rr, rowok, ok);
                    if (rdC) { v.x += gp.beta * creg[g][rr].x; v.y += gp.beta * creg[g][rr].y; }
                    if (ok) *reinterpret_cast<__attribute__((address_space(1))) doublex2 *>(reinterpret_cast<uintptr_t>(p)) = v;
                    // the store is issued iff the row exists (lane 0's columns always do).  Counting one that is NOT issued would make the
                    // landing waits of pump() too weak; the count has to be exact or low
                    if (rowok) n_ops = __builtin_amdgcn_readfirstlane(n_ops + 1);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the staged rows are in registers: the buffer may be rewritten
            if (lane == 0) WS_ST(&ctl->sfree[b][h], hs + 1);
        }
        if (dead) break;
        if (rdC && it + 1 < ntile) load_c(it + 1);
    }
    // (every chunk has been issued and published by now: the last tile's groups cannot be staged before its last chunk was consumed)
    if (ws_diag && lane == 0) { gp.stamps[42] = t_vm; gp.stamps[43] = n_pump; gp.stamps[44] = n_issue; gp.stamps[45] = __builtin_amdgcn_s_memtime() - t_begin; gp.stamps[48] = t_issue; }
    if (dead && lane == 0) { WS_ST(&ctl->abort, 1); atomicAdd(&g_ws_aborts, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool AKM>
__global__ __launch_bounds__(512, 2) void dgemm_ws_kernel(DgemmParams gp)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char wsm[];
    WsCtl *const ctl = reinterpret_cast<WsCtl *>(wsm + (WRS + WSG) * WS_SLOT);
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)(sizeof(WsCtl) / sizeof(int)); i += 512) reinterpret_cast<int *>(ctl)[i] = 0;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(tid >> 6) < 4) ws_consumer<AKM>(gp, wsm);
    else ws_helper<AKM>(gp, wsm);
}

