// Persistent form of the fp32 implicit-GEMM convolution (v_mfma_f32_32x32x2_f32), round 4.
//
// Same fused op and the same arithmetic as conv_f32.hip (reference: core/parse_model.py:27-52,72,134,155-156): every output is
// the same k-ordered sum, so results are BIT-IDENTICAL to the classic tiles (tests/test_gpu_parity.py::test_conv_every_tile_shape
// and test_persistent_tiles_bit_identical_to_classic).  What changes is who pays for what outside the K loop:
//   * a launch has as many workgroups as the chip holds at once (occupancy x CUs).  Workgroup w OWNS positions w, w + G, ... of
//     the classic tile order for the whole rounds of the launch (the workgroups running at any moment still share activation
//     rows / weight rows in one L2); the fewer-than-G positions left over are PULLED by whoever gets there first, from one cursor
//     per XCD (a workgroup whose XCD has run dry takes from the others), so that a launch of 7.04 rounds is not paid as 8
//     (profiles/r04_tile_sweep_f32_persistent_static_b64_s416.txt: the all-static first build, -4 % against the classic tiles on the
//     large layers; pulling EVERY tile, the second build, cost more than it balanced: r04_tile_sweep_f32_persistent_pull_all_*.txt,
//     1x1 layers 110 -> 65 TFLOP/s -- the returning atomic sits in the wave's memory queue in front of the next K tile).  A pull
//     is one relaxed agent-scope atomic add by one lane, issued a tile ahead of its use and handed to the other waves through an
//     LDS word behind a barrier the K loop has anyway; nobody waits for anybody (no spin).  The last workgroup to leave zeroes the
//     cursors for the next launch (cdna_hip_programming.md Guideline 16: counters, not flags);
//   * the operand ring (two LDS stages, direct-to-LDS buffer loads) runs ACROSS tile boundaries: the first K tile of the next
//     output tile is requested at the top of the current tile's last K iteration, so it lands under that iteration's MFMAs and
//     the epilogue; no wave ever sits in "first fetch + barrier";
//   * the per-tile row state comes from a table built at plan time (int2 per output row: pixel index of tap (0,0) + the 9-bit
//     tap-validity mask; for concat convs the two source pixel indices) instead of ~260 vector instructions of pixel
//     decomposition per workgroup -- the fp32 MFMA shares the SIMD with vector ALU work, every vector instruction outside the
//     K loop is matrix-pipe time (profiles/r03_f32_phase_stamps.txt: 20 + 13 us of a 132-us workgroup life); the table entries of
//     the next tile are requested at the start of the current one;
//   * kernel arguments, buffer descriptors, fragment addresses, weight-row offsets: once per workgroup, not once per tile.
#include <algorithm>
#include <type_traits>

#include "y3_kernels.h"

namespace y3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int PBK = 32;   // K tile (floats): one 128-B LDS row per operand row

// MODE: 0 = single source, row state from the table (3x3 convs, stride 1 or 2)
//       1 = single source 1x1 / stride 1: row m IS pixel m (no table)
//       2 = 1x1 over (optionally x2-upsampled) src0 (+) src1: the table holds the two source pixel indices
template <int TM, int TN, int WR, int WC, int MODE, int MINW>
__global__ __launch_bounds__(64 * WR * WC, MINW) void conv_f32_pers(const ConvArgs p)
{
    constexpr int BM = 32 * TM * WR;
    constexpr int BN = 32 * TN * WC;
    constexpr int NT = 64 * WR * WC;
    constexpr int RP = NT / 8;   // rows per load pass (8 lanes x 16 B cover one 32-float row)
    constexpr int AP = BM / RP, BP = BN / RP;
    static_assert(BM % RP == 0 && BN % RP == 0 && AP >= 1 && BP >= 1, "tile too small for the thread count");
    constexpr int STAGE = (BM + BN) * PBK;   // floats per ring stage
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int fr = lane & 31, fh = lane >> 5;

    // ---- positions 0 .. T-1 of the classic launch order: position l belongs to XCD l & 7 ---------------------------------
    const int tilesN = p.CoutPad / BN;
    const int tilesM = (p.M + BM - 1) / BM;
    const int KT = p.K / PBK;
    int gn = 0, gm = 0, nb_x = 0, rows_x = 0, T;
    if (p.xcd_gn > 0) {   // XCD-blocked order (see conv_f32.hip): position l -> XCD l & 7 owns an M block x N block
        gn = p.xcd_gn;
        gm = 8 / gn;
        nb_x = tilesN / gn;
        for (int xm = 0; xm < gm; ++xm) rows_x = max(rows_x, (xm + 1) * tilesM / gm - xm * tilesM / gm);
        T = 8 * rows_x * nb_x;
    } else {
        T = tilesM * tilesN;
    }
    // position -> (m0, n0); false for the padding positions of an uneven blocked split
    auto coords = [&](int l, int &m0, int &n0) -> bool {
        const int xcd = l & 7, j = l >> 3;
        int mt, nt;
        if (gn > 0) {
            const int xm = xcd / gn, xn = xcd - xm * gn;
            const int mlo = xm * tilesM / gm, mhi = (xm + 1) * tilesM / gm;
            const int lm = j / nb_x;
            mt = mlo + lm;
            nt = xn * nb_x + (j - lm * nb_x);
            if (mt >= mhi) return false;
        } else {
            const int q8 = T >> 3, r8 = T & 7;
            const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
            mt = logical / tilesN;
            nt = logical - mt * tilesN;
        }
        m0 = mt * BM;
        n0 = nt * BN;
        return true;
    };
    // ---- who runs which position ------------------------------------------------------------------------------------------
    // Workgroup w OWNS positions w, w + G, ... of the first RS whole rounds (G = gridDim.x, a multiple of 8: all of them in one
    // XCD's share); the T - RS * G positions left over (fewer than G) are PULLED by whoever gets there: one cursor per XCD counts
    // the left-over positions of that XCD's share that have been handed out, a workgroup whose XCD has run dry takes from the
    // others.  A launch of fewer than two whole rounds (and every launch when p.pers_ctr is null) is all static.
    int *const ctr = p.pers_ctr;                                                   // [0..7] one cursor per XCD, [8] workgroups that have left
    const int G = (int)gridDim.x;
    const bool dynamic = ctr != nullptr && T >= 2 * G && T % G != 0;
    const int RS = dynamic ? T / G : (T + G - 1) / G;                              // static rounds
    const int S_lim = dynamic ? RS * G : T;                                        // positions below S_lim are owned, the others pulled
    const int jbase = (RS * G) >> 3;                                               // left-over position c of XCD x: x + 8 * (jbase + c)
    const int xcd = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7);         // HW_REG_XCC_ID: which cursor to pull from first (speed only)
    auto lshare = [&](int x) { return (gn > 0 ? rows_x * nb_x : (T >> 3) + (x < (T & 7) ? 1 : 0)) - jbase; };   // left-over positions of XCD x
    // wave 0 only, dynamic launches only.  A left-over position for this workgroup, or T when they have all been handed out.
    // One 32-byte look at the eight cursors tells which XCDs still have any; the atomic add that follows can still lose a race.
    auto pull_sync = [&]() -> int {
        for (;;) {
            int c = 0x7fffffff;
            if (lane < 8) c = __hip_atomic_load(ctr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned avail = (unsigned)__builtin_amdgcn_ballot_w64(lane < 8 && c < lshare(lane)) & 0xffu;
            if (!avail) return T;
            const unsigned rot = ((avail >> xcd) | (avail << (8 - xcd))) & 0xffu;   // bit k: XCD (xcd + k) & 7
            const int x = (xcd + __builtin_ctz(rot)) & 7;
            int j = 0;
            if (lane == 0) j = __hip_atomic_fetch_add(ctr + x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            j = __builtin_amdgcn_readfirstlane(j);
            int mm, nn;
            if (j < lshare(x) && coords(x + 8 * (jbase + j), mm, nn)) return x + 8 * (jbase + j);   // (padding positions of a blocked split are skipped)
        }
    };
    auto leave = [&]() {   // dynamic launches: every workgroup, once, after its last pull; the last one to leave zeroes the cursors for the next launch
        if (dynamic && tid == 0) {
            const int old = __hip_atomic_fetch_add(ctr + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == G - 1) {
                for (int x = 0; x < 9; ++x) __hip_atomic_store(ctr + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto next_owned = [&](int l) -> int {   // first valid owned position >= l on this workgroup's walk, or T
        int mm, nn;
        while (l < S_lim && !coords(l, mm, nn)) l += G;
        return l < S_lim ? l : T;
    };
    int *const slot = reinterpret_cast<int *>(smem + 2 * STAGE);   // two words behind the ring: pulled positions handed from wave 0 to the others
    int pos = next_owned((int)blockIdx.x);
    if (pos >= T) {   // (more workgroups than tiles, or only padding positions: possible in all-static launches only)
        leave();
        return;
    }
    int posn = next_owned(pos + G);   // dynamic launches have RS >= 2: the second tile is owned as well (blocked splits may lack it; they then stop early)
    int m0 = 0, n0 = 0;
    coords(pos, m0, n0);

    // ---- per-workgroup constants -------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.src0), 0, p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(MODE == 2 ? p.src1 : p.src0), 0, MODE == 2 ? p.src1_bytes : p.src0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wpk), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(p.dst, 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(p.residual ? p.residual : p.dst), 0, p.dst_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(MODE == 1 ? p.src0 : static_cast<const void *>(p.rowtab)), 0, MODE == 1 ? 0u : p.rowtab_bytes, 0x00020000);
    // a voffset equal to num_records fails the buffer's bounds check: the load returns 0 (the SCALAR offset takes no part in that check)
    const unsigned OOB0 = p.src0_bytes;

    const int lrow = tid >> 3;
    // the 16-B piece that lands in physical chunk tid & 7 of an LDS row is logical chunk (tid & 7) ^ ((row >> 1) & 7)
    const int lchunk = (((tid & 7) ^ ((lrow >> 1) & 7)) * 4);
    unsigned boff[BP];   // byte offset of this lane's piece of weight row (j * RP + lrow), k = 0; the tile's n0 * K * 4 rides in the soffset
#pragma unroll
    for (int j = 0; j < BP; ++j) boff[j] = (unsigned)((j * RP + lrow) * p.K + lchunk) * 4u;
    const int C1 = p.Cin - p.C0;
    const int taps = p.ksize * p.ksize;
    // K order (see conv_f32.hip): tap-major, or chunk-major with CK channels per chunk for 3x3 convs
    const int CK = (MODE == 0 && p.k_chunk > 0 && p.k_chunk < p.Cin) ? p.k_chunk : p.Cin;

    const int a_frag = (wr * 32 * TM + fr) * PBK;
    const int b_frag = BM * PBK + (wc * 32 * TN + fr) * PBK;
    int foff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) foff[q] = (((2 * q + fh) ^ ((fr >> 1) & 7)) * 4);

    // ---- row state of the tile being FETCHED ------------------------------------------------------------------------------
    unsigned avoff[AP];                       // voffset of this lane's piece for the current tap (or the out-of-range sentinel)
    unsigned okmask[MODE == 0 ? AP : 1];      // bit t: tap t of the row reads inside the image
    unsigned abase4[MODE == 0 ? AP : 1];      // byte offset of the lane's piece at tap (0,0), channel 0
    unsigned avoff1[MODE == 2 ? AP : 1];
    u32x2 rinfo[MODE == 1 ? 1 : AP];          // table entries in flight (requested one tile ahead)
    int kglob = 0, tap = 0, c0 = 0, cend = CK, nK4 = 0;

    auto request_rows = [&](int m0_) {
        if (MODE != 1) {
#pragma unroll
            for (int i = 0; i < AP; ++i)   // rows >= M: beyond num_records -> (0, 0) = no valid tap / pixel 0 (results dropped by the epilogue)
                rinfo[i] = __builtin_amdgcn_raw_buffer_load_b64(rst, (int)((unsigned)(m0_ + i * RP + lrow) * 8u), 0, 0);
        }
    };
    auto set_tap = [&]() {
        if (MODE == 0) {
            const int u = tap / p.ksize, v = tap - u * p.ksize;
            const unsigned toff4 = (unsigned)((u * p.W + v) * p.Cin) * 4u;
#pragma unroll
            for (int i = 0; i < AP; ++i) avoff[i] = ((okmask[i] >> tap) & 1u) ? abase4[i] + toff4 : OOB0;
        }
    };
    auto apply_rows = [&](int m0_, int n0_) {   // the requested entries have landed: they become the fetch state
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                abase4[i] = (rinfo[i][0] * (unsigned)p.Cin + (unsigned)lchunk) * 4u;
                okmask[i] = rinfo[i][1];
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const unsigned m = (unsigned)(m0_ + i * RP + lrow);
                avoff[i] = m < (unsigned)p.M ? (m * (unsigned)p.Cin + (unsigned)lchunk) * 4u : OOB0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                avoff[i] = (rinfo[i][0] * (unsigned)p.C0 + (unsigned)lchunk) * 4u;
                avoff1[i] = (rinfo[i][1] * (unsigned)C1 + (unsigned)lchunk) * 4u;
            }
        }
        kglob = 0;
        tap = 0;
        c0 = 0;
        cend = CK;
        // (readfirstlane: the tile position has been through an LDS word and loop-carried selects; hipcc must be able to PROVE the
        // scalar offset of the buffer loads wave-uniform or it wraps every one of them in a waterfall loop -- cdna_hip_programming.md T20)
        nK4 = __builtin_amdgcn_readfirstlane(n0_ * p.K * 4);
        set_tap();
    };

    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto fetch = [&](int buf) {   // one K tile of both operands into ring stage buf; advances the K walk
        float *sa = smem + buf * STAGE + wave * 8 * PBK;   // wave w fills rows [pass * RP + 8 w, +8)
        float *sb = sa + BM * PBK;
        if (MODE == 2 && c0 >= p.C0) {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_ptr)(sa + i * RP * PBK), 16, (int)avoff1[i], (c0 - p.C0) * 4, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < AP; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lds_ptr)(sa + i * RP * PBK), 16, (int)avoff[i], c0 * 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(sb + j * RP * PBK), 16, (int)boff[j], nK4 + kglob * 4, 0, 0);
        c0 += PBK;
        if (c0 == cend) {
            ++tap;
            if (MODE == 0 && tap == taps && cend != p.Cin) {   // chunk-major order: next channel chunk, first tap again
                tap = 0;
                cend += CK;
            }
            c0 = cend - CK;
            set_tap();
        }
        kglob = tap * p.Cin + c0;
    };

    // ---- start stagger ----------------------------------------------------------------------------------------------------
    // The R workgroups of a CU start together and run equal work: left alone they reach their epilogues (and their first-K-tile
    // waits) together, launch after launch, and the matrix pipe idles through all of them at once -- the classic launch does not
    // have this problem (its workgroups are dispatched whenever a slot frees up).  Group r of R (workgroups r * G / R ...: the
    // dispatcher deals the first G / R workgroups one to a CU, then the next G / R) sleeps r x pers_stagger % of the matrix time of
    // one tile before its first fetch, so that the CU's workgroups run out of phase.  Speed only.
    if (p.pers_stagger > 0 && p.pers_groups > 1 && G % p.pers_groups == 0) {
        const int grp = (int)blockIdx.x / (G / p.pers_groups);
        const long long cyc = (long long)grp * KT * 32 * TM * TN * 64 * p.pers_stagger / 100;
        for (long long c = 0; c < cyc; c += 127 * 64) __builtin_amdgcn_s_sleep(127);
    }

    // ---- first tile of this workgroup -------------------------------------------------------------------------------------
    request_rows(m0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    apply_rows(m0, n0);
    int stage = 0;
    fetch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int row_bytes = p.Cout * 4;
    for (int it = 0;; ++it) {
        int m0n = 0, n0n = 0;
        const bool has_next = posn < T;
        // the position AFTER the next one: owned (arithmetic), or -- past this workgroup's owned rounds -- pulled: the atomic is
        // issued here by one lane, resolved by wave 0 in the last K iteration and read by everybody after the epilogue
        int posnn = T, pend = 0;
        bool pulled = false;
        if (has_next) {
            coords(posn, m0n, n0n);
            m0n = __builtin_amdgcn_readfirstlane(m0n);
            n0n = __builtin_amdgcn_readfirstlane(n0n);
            if (posn < S_lim) posnn = next_owned(posn + G);
            pulled = dynamic && posnn >= T;
            // The table entries of the next tile (and the pull) are requested at the END of the first K iteration, behind that
            // iteration's operand fetch: vector-memory operations complete in issue order, so requested at the top they would sit in
            // front of the K tile the iteration waits for -- a cold read (and a returning atomic) in front of an L2 hit, once per tile.
            if (KT == 1) {
                request_rows(m0n);
                if (pulled && wave == 0 && lane == 0) pend = __hip_atomic_fetch_add(ctr + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) {
                fetch(stage ^ 1);
            } else if (has_next) {
                // last K tile of this output tile: the ring moves on to the NEXT tile's first K tile
                if (KT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the table entries were requested in this very iteration)
                apply_rows(m0n, n0n);
                fetch(stage ^ 1);
                if (pulled && wave == 0) {   // the pull issued at the top of this tile has returned: resolve it and hand it over
                    const int j = __builtin_amdgcn_readfirstlane(pend);
                    int mm, nn, q = xcd + 8 * (jbase + j);
                    if (j >= lshare(xcd) || !coords(q, mm, nn)) q = pull_sync();
                    if (lane == 0) slot[it & 1] = q;   // read by every wave after this iteration's barrier; rewritten two tiles on at the earliest
                }
            }
            // (two copies of the block, one per ring stage: with the stage in a register every fragment address costs a vector add)
            auto mma = [&](auto stage_tag) {
                constexpr int ST = decltype(stage_tag)::value;
                const float *sa = smem + ST * STAGE + a_frag;
                const float *sb = smem + ST * STAGE + b_frag;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4 *>(sa + i * 32 * PBK + foff[q]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4 *>(sb + j * 32 * PBK + foff[q]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
                }
            };
            if (stage == 0) mma(std::integral_constant<int, 0>{}); else mma(std::integral_constant<int, 1>{});
            // the K tile requested at the top has landed -- for every wave, after the barrier -- and every wave is done reading this
            // stage (its fragment reads were consumed by its MFMAs).  Raw s_barrier + explicit counts: __syncthreads() would drain
            // the memory queue, and the first iteration leaves the table entries / the pull of the next tile in flight.
            constexpr int NTAB = MODE == 1 ? 0 : AP;
            if (kt == 0 && has_next && KT > 1) {
                request_rows(m0n);
                if (pulled && wave == 0) {
                    if (lane == 0) pend = __hip_atomic_fetch_add(ctr + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NTAB + 1) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NTAB) : "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            stage ^= 1;
        }

        // ---- epilogue of (m0, n0): identical arithmetic to conv_f32.hip ---------------------------------------------------
        // accumulator element e of lane l: column (n) = l & 31, row (m) = (e & 3) + 8*(e >> 2) + 4*(l >> 5)
        const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.Cout);
        auto emit = [&](auto leaky_tag, auto res_tag, auto interior_tag) {
            constexpr bool LEAKY = decltype(leaky_tag)::value, RES = decltype(res_tag)::value;
            constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + (wc * TN + j) * 32 + fr;
                const float sh = p.shift[n];   // the BN scale is folded into the packed weights (y3_api.cpp)
                const bool n_ok = n < p.Cout;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mbase = m0 + (wr * TM + i) * 32 + 4 * fh;
                    const unsigned vbase = (unsigned)(mbase * p.Cout + n) * 4u;
                    unsigned off[16];
                    if (!INTERIOR) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int m = mbase + (e & 3) + 8 * (e >> 2);
                            off[e] = (n_ok && m < p.M) ? (unsigned)(m * p.Cout + n) * 4u : p.dst_bytes;
                        }
                    }
                    float r[16];
                    if (RES) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int so = ((e & 3) + 8 * (e >> 2)) * row_bytes;
                            r[e] = __builtin_bit_cast(float, INTERIOR ? __builtin_amdgcn_raw_buffer_load_b32(rsr, (int)vbase, so, 0)
                                                                      : __builtin_amdgcn_raw_buffer_load_b32(rsr, (int)off[e], 0, 0));
                        }
                    }
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        f32x2 v2 = f32x2{acc[i][j][e], acc[i][j][e + 1]} + f32x2{sh, sh};
                        if (LEAKY) {
                            const f32x2 t2 = v2 * f32x2{0.1f, 0.1f};
                            v2 = f32x2{fmaxf(v2[0], t2[0]), fmaxf(v2[1], t2[1])};   // == (v >= 0 ? v : 0.1 v) for every finite v
                        }
                        if (RES) v2 = f32x2{r[e], r[e + 1]} + v2;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int ee = e + h;
                            const int so = ((ee & 3) + 8 * (ee >> 2)) * row_bytes;
                            if (INTERIOR)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v2[h]), rsd, (int)vbase, so, 0);
                            else
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v2[h]), rsd, (int)off[ee], 0, 0);
                        }
                    }
                }
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        auto emit2 = [&](auto leaky_tag, auto res_tag) {
            if (interior) emit(leaky_tag, res_tag, T_{}); else emit(leaky_tag, res_tag, F_{});
        };
        if (p.residual) {
            if (p.leaky) emit2(T_{}, T_{}); else emit2(F_{}, T_{});
        } else {
            if (p.leaky) emit2(T_{}, F_{}); else emit2(F_{}, F_{});
        }

        if (!has_next) break;
        pos = posn;
        posn = __builtin_amdgcn_readfirstlane(pulled ? slot[it & 1] : posnn);
        m0 = m0n;
        n0 = n0n;
    }
    leave();
}

// resident workgroups per CU of an instantiation on a device (occupancy query once per (instantiation, device))
struct ResidentOnce { int per_cu[64] = {0}; int cus[64] = {0}; };

template <int TM, int TN, int WR, int WC, int MODE, int MINW>
hipError_t launch_pers(const ConvArgs &a, hipStream_t s)
{
    constexpr int BM = 32 * TM * WR, BN = 32 * TN * WC, NT = 64 * WR * WC;
    const size_t lds = 2 * (size_t)(BM + BN) * PBK * sizeof(float) + 16;   // the ring + the two hand-over words
    auto k = conv_f32_pers<TM, TN, WR, WC, MODE, MINW>;
    static LdsAttrOnce attr;
    if (hipError_t e = set_max_lds_once(attr, reinterpret_cast<const void *>(k), (int)lds); e != hipSuccess) return e;
    static ResidentOnce res;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!res.per_cu[dev]) {
        int occ = 0, cus = 0;
        if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, NT, lds); e != hipSuccess) return e;
        if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return e;
        if (occ < 1 || cus < 8) return hipErrorInvalidValue;
        res.per_cu[dev] = occ;
        res.cus[dev] = cus;
    }
    const int tilesM = (a.M + BM - 1) / BM, tilesN = a.CoutPad / BN;
    int T = tilesM * tilesN;
    if (a.xcd_gn > 0) {
        if (8 % a.xcd_gn || tilesN % a.xcd_gn) return hipErrorInvalidValue;
        const int gm = 8 / a.xcd_gn;
        int rows = 0;
        for (int xm = 0; xm < gm; ++xm) rows = std::max(rows, (xm + 1) * tilesM / gm - xm * tilesM / gm);
        T = 8 * rows * (tilesN / a.xcd_gn);
    }
    int per_cu = res.per_cu[dev];
    if (a.pers_wg_per_cu > 0 && a.pers_wg_per_cu < per_cu) per_cu = a.pers_wg_per_cu;
    int grid = per_cu * res.cus[dev];
    grid -= grid % 8;
    if (grid > T) grid = (T + 7) / 8 * 8;   // fewer tiles than slots: about one tile each
    if (MODE != 1 && (!a.rowtab || a.rowtab_bytes < (unsigned)a.M * 8u)) return hipErrorInvalidValue;
    if (!a.pers_ctr) return hipErrorInvalidValue;
    ConvArgs b = a;
    b.pers_groups = (grid == per_cu * res.cus[dev]) ? per_cu : 1;   // a full grid: per_cu workgroups on every CU
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, s, b);
    return hipGetLastError();
}

template <int TM, int TN, int WR, int WC, int MINW>
hipError_t launch_pers_mode(const ConvArgs &a, hipStream_t s)
{
    if (a.src1) return launch_pers<TM, TN, WR, WC, 2, MINW>(a, s);
    if (a.ksize == 1 && a.stride == 1) return launch_pers<TM, TN, WR, WC, 1, MINW>(a, s);
    return launch_pers<TM, TN, WR, WC, 0, MINW>(a, s);
}
}  // namespace

hipError_t launch_conv_f32_pers(const ConvArgs &a, int tile, hipStream_t s)
{
    switch (tile) {
        case 33: return launch_pers_mode<1, 2, 2, 2, 3>(a, s);   // 64x128, 4 waves: 48 KB of ring -> 3 workgroups per CU
        case 34: return launch_pers_mode<1, 1, 2, 2, 4>(a, s);   // 64x64, 4 waves: 32 KB -> 4 workgroups per CU (register budget of 4 waves per SIMD)
        case 35: return launch_pers_mode<1, 1, 4, 2, 3>(a, s);   // 128x64, 8 waves: 48 KB -> 3 workgroups = 6 waves per SIMD
        case 36: return launch_pers_mode<2, 2, 2, 2, 2>(a, s);   // 128x128, 4 waves (64x64 wave tiles): 64 KB -> 2 workgroups per CU
        case 37: return launch_pers_mode<1, 2, 4, 2, 4>(a, s);   // 128x128, 8 waves (32x64 wave tiles): 64 KB -> 2 workgroups = 4 waves per SIMD
        default: return hipErrorInvalidValue;
    }
}

}  // namespace y3
