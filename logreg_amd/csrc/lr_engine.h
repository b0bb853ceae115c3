// lr_engine.h -- the host side between the C ABI (lr_api.hip) and the kernel launches: packs the by-value kernel argument structs
// from a model handle, a plan (lr_plan.h) and the caller's options, owns the stepwise engine's workspace and drives its launches
// (load -> partial -> update per evaluation; the interior leapfrog steps on the reduced-precision kernels where the policy allows).
// No arithmetic of the path happens here.  Included once, by lr_api.hip (after lr_model.h and lr_plan.h).
#pragma once

namespace {

template <typename T, int P> lr::ModelArgs<T, P> model_args(const lr_model* m) {
    lr::ModelArgs<T, P> a{};
    a.rows = static_cast<const T*>(m->d_rows);
    a.rows_tw = static_cast<const float*>(m->d_rows_tw);
    a.rows_mf = static_cast<const float*>(m->d_xmf);
    a.ops_mf = static_cast<const unsigned char*>(m->d_xms);
    a.n = m->n;
    for (int j = 0; j < P; ++j) a.prior.inv_var[j] = (T)m->inv_var[j];
    a.prior.lprior_const = m->lprior_const;
    return a;
}

// the chain count every chain-count-dependent choice is made for (lr_run_opts.plan_chains: a shard plans as the whole run)
int64_t plan_count(const lr_run_opts* o) { return o->plan_chains > 0 ? (int64_t)o->plan_chains : o->n_chains; }

struct RunSpec {
    int kind;
    double step;
    int l;
    double a[kMaxP], b[kMaxP], c[kMaxP];
};

template <typename T, int P>
int do_eval_t(lr_model* m, const Plan& pl, hipStream_t st, int64_t C, const void* beta, void* ll, void* lprior,
              void* lpost, void* grad) {
    auto ma = model_args<T, P>(m);
    lr::EvalArgs<T> ea;
    ea.beta = static_cast<const T*>(beta);
    ea.C = C;
    ea.p = m->p;
    ea.ll = static_cast<T*>(ll);
    ea.lprior = static_cast<T*>(lprior);
    ea.lpost = static_cast<T*>(lpost);
    ea.grad = static_cast<T*>(grad);
    lr::LaunchCfg cfg{pl.mode, pl.G, pl.R, 0, st, pl.lds_bytes, m->dbg.residency_cap ? m->cus : 0};
    const int rc = m->table->launch_eval(&cfg, C, &ma, &ea);
    if (rc != 0) return fail(rc == -3 ? LR_ERR_UNSUPPORTED : LR_ERR_HIP, "eval launch failed (%d): %s", rc,
                             hipGetErrorString(hipGetLastError()));
    return LR_OK;
}

template <typename T, int P>
int do_chain_t(lr_model* m, const Plan& pl, hipStream_t st, const RunSpec& rs, const lr_run_opts* o, void* state,
               double* lp_state, void* out, uint32_t* accepts) {
    auto ma = model_args<T, P>(m);
    lr::ChainArgs<T, P> ca{};
    ca.state = static_cast<T*>(state);
    ca.lp_state = lp_state;
    ca.out = static_cast<T*>(out);
    ca.accepts = accepts;
    ca.C = o->n_chains;
    ca.chain_offset = o->chain_offset;
    ca.iters = o->iters;
    ca.thin = o->thin;
    ca.iter_offset = o->iter_offset;
    ca.seed = o->seed;
    ca.p = m->p;
    ca.l = rs.l;
    ca.step = (T)rs.step;
    for (int j = 0; j < P; ++j) {
        ca.a[j] = (T)rs.a[j];
        ca.b[j] = (T)rs.b[j];
        ca.c[j] = (T)rs.c[j];
        const double ks = sizeof(T) == 4 ? 1.4426950408889634 : 1.0;  // lr::ExpScale<T>::k
        ca.d[j] = (T)(rs.b[j] * ks);
        ca.e[j] = (T)(m->inv_var[j] / ks);
    }
    ca.stats = lr::StatsArgs{o->stats, o->stats_batch, o->stats_first};
    ca.interior_bf16 = rs.kind == lr::KIND_HMC && pl.mode == lr::MODE_MFMA && o->precision != LR_PREC_FULL;
    // a plan in two parts: chains whose position in the PLANNED run is below pl.split go to (G, R), the rest to (G2, R2)
    const int64_t C = o->n_chains, pos0 = o->plan_chains > 0 ? o->chain_offset - o->plan_first : 0;
    const int64_t head = pl.split > 0 ? (pl.split - pos0 < 0 ? 0 : (pl.split - pos0 > C ? C : pl.split - pos0)) : C;
    // Both parts present: the remainder runs CO-RESIDENT with the head, on the handle's side stream -- fork from the caller's
    // stream (everything enqueued so far), join back into it.  The head is VALU-issue-bound on one wave per SIMD, the remainder
    // (wide lane groups) mostly waits on cross-lane reductions: together 0.542 instead of 0.639 ms per step at 5120 chains
    // (profiles/r4_two_part_corun.txt, r4_two_part_check.txt).  The head keeps its residency cap (one workgroup per CU); the
    // remainder asks for none, so that it fits beside the head.
    // ... while the head is at most two waves per SIMD: beside a head of three the remainder only gets in the way (HMC, 13 312 chains:
    // 1.32 ms back to back, 1.47 co-resident, 1.40 as one launch; profiles/r4_chain_grid_corun_all.txt) -- then the parts run in turn
    const bool two = head > 0 && head < C;
    // (the threaded-ll kernels gain from the overlap at three waves as well: MALA +6 %, RWMH +10 % co-resident against -2 % / -1 % in turn)
    // (a matrix-core head: workgroups per CU instead -- 16 chains per workgroup at S >= 4, 64 at S = 1)
    const int64_t head_rounds = pl.mode == lr::MODE_MFMA ? head / ((pl.G == 1 ? 64 : 16) * (int64_t)m->cus) : head * pl.G / 64 / (4LL * m->cus);
    const bool both = two && pl.corun && (rs.kind != lr::KIND_HMC || head_rounds <= 2);
    lr_model::Side* side_slot = nullptr;
    if (both) {
        for (auto& e : m->sides)
            if (e.caller == st) side_slot = &e;
        if (!side_slot) {  // created all or nothing: a partly made slot is destroyed again, never kept
            lr_model::Side e{st, nullptr, nullptr, nullptr};
            const bool ok = hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking) == hipSuccess &&
                            hipEventCreateWithFlags(&e.ev_fork, hipEventDisableTiming) == hipSuccess &&
                            hipEventCreateWithFlags(&e.ev_join, hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                const hipError_t err = hipGetLastError();
                if (e.ev_join) (void)hipEventDestroy(e.ev_join);
                if (e.ev_fork) (void)hipEventDestroy(e.ev_fork);
                if (e.stream) (void)hipStreamDestroy(e.stream);
                return fail(LR_ERR_HIP, "creating the side stream of a two-part launch failed: %s", hipGetErrorString(err));
            }
            m->sides.push_back(e);
            side_slot = &m->sides.back();
        }
        if (hipEventRecord(side_slot->ev_fork, st) != hipSuccess || hipStreamWaitEvent(side_slot->stream, side_slot->ev_fork, 0) != hipSuccess)
            return fail(LR_ERR_HIP, "forking the two-part launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    for (int part = 0; part < 2; ++part) {
        ca.first = part == 0 ? 0 : head;
        ca.count = part == 0 ? head : C - head;
        if (ca.count <= 0) continue;
        const bool side = both && part == 1;
        lr::LaunchCfg cfg{part == 0 ? pl.mode : pl.mode2, part == 0 ? pl.G : pl.G2, part == 0 ? pl.R : pl.R2, rs.kind, side ? side_slot->stream : st,
                          part == 0 || pl.mode2 == lr::MODE_MIXED ? pl.lds_bytes : 0, m->dbg.residency_cap && !side ? m->cus : 0};
        const int rc = m->table->launch_chain(&cfg, ca.count, &ma, &ca);
        if (rc != 0) return fail(rc == -3 ? LR_ERR_UNSUPPORTED : LR_ERR_HIP, "chain launch failed (%d): %s", rc,
                                 hipGetErrorString(hipGetLastError()));
    }
    if (both) {
        if (hipEventRecord(side_slot->ev_join, side_slot->stream) != hipSuccess || hipStreamWaitEvent(st, side_slot->ev_join, 0) != hipSuccess)
            return fail(LR_ERR_HIP, "joining the two-part launch failed: %s", hipGetErrorString(hipGetLastError()));
    }
    return LR_OK;
}

template <typename T, int P>
int setup_tall(lr_model* m, const Plan& pl, hipStream_t st, int64_t C, int64_t Cp, lr::TallArgs<T, P>* pa) {
    // C: chains of this call (sizes the workspace and the grids);  Cp: chains the slicing decisions are made for
    lr::TallArgs<T, P>& a = *pa;
    // every field starts from zero: the flags a run sets only on some paths (part_f32, fuse_mid, interior) were left to whatever the
    // caller's stack held -- found in round 5 when a float64 run on the trajectory kernels, which never touches part_f32, now and then
    // had its end-point update read float64 partials as float32
    a = lr::TallArgs<T, P>{};
    const int RS = pl.G;
    auto align = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t vec = align((size_t)C * P * sizeof(T)), dbl = align((size_t)C * sizeof(double));
    // row slices of the reduced-precision interior leapfrog steps (lr_plan.h)
    const InteriorPlan ip = plan_interior(m, Cp);
    const int RS_i = ip.RS_i, rs_waves = ip.waves;
    const int64_t slice_len_i = ip.slice_len_i;
    const int RSmax = RS_i > RS ? RS_i : RS;
    const size_t pg = align((size_t)RSmax * C * P * sizeof(T)), cv = align((size_t)2 * P * sizeof(T));
    // (second state pair + second partial buffer + constants: the fused interior steps of the row-split kernels)
    const size_t need = 4 * vec + 2 * dbl + align((size_t)C * 4) + pg + align((size_t)RS * C * sizeof(double)) +
                        (RS_i > 0 && (m->P > 32 || rs_waves == 16) ? 2 * vec + pg : 0) + cv;
    lr_model::Ws* slot = nullptr;
    for (auto& e : m->ws)
        if (e.stream == st) slot = &e;
    if (!slot) {
        if (m->ws.size() >= 16) {  // a caller cycling through streams: drop every workspace once all work is done
            if (hipDeviceSynchronize() != hipSuccess) return fail(LR_ERR_HIP, "hipDeviceSynchronize failed");
            for (auto& e : m->ws)
                if (e.p) (void)hipFree(e.p);
            m->ws.clear();
        }
        m->ws.push_back({st, nullptr, 0});
        slot = &m->ws.back();
    }
    if (need > slot->bytes) {
        if (slot->p) {
            // work already enqueued on this stream may still use the old block: hipFree waits for the device
            (void)hipFree(slot->p);
        }
        slot->p = nullptr;
        slot->bytes = 0;
        if (hipMalloc(&slot->p, need) != hipSuccess) return fail(LR_ERR_NOMEM, "stepwise workspace of %zu bytes", need);
        slot->bytes = need;
    }
    unsigned char* w = static_cast<unsigned char*>(slot->p);
    auto carve = [&](size_t b) { unsigned char* r = w; w += b; return r; };
    std::memset(&a, 0, sizeof(a));
    a.rows = static_cast<const T*>(m->d_rows);
    a.rows_tw = static_cast<const float*>(m->d_rows_tw);
    a.n = m->n;
    a.slice_len = pl.R;
    a.RS = RS;
    for (int j = 0; j < P; ++j) a.prior.inv_var[j] = (T)m->inv_var[j];
    a.prior.lprior_const = m->lprior_const;
    a.x = (T*)carve(vec);
    a.g = (T*)carve(vec);
    a.q1 = (T*)carve(vec);
    a.pm = (T*)carve(vec);
    a.lp = (double*)carve(dbl);
    a.aux = (double*)carve(dbl);
    a.nacc = (uint32_t*)carve(align((size_t)C * 4));
    a.part_g = (T*)carve(pg);
    a.RS_i = RS_i;
    a.traj_tiles = 1, a.traj_fmt = 0;  // (set per run by do_stepwise_t)
    a.slice_len_i = slice_len_i;
    a.rowsplit_waves = rs_waves;
    a.part_v = (double*)carve(align((size_t)RS * C * sizeof(double)));
    a.cvec = (const T*)carve(cv);
    if (RS_i > 0 && (m->P > 32 || rs_waves == 16)) {  // alternates, parked in the *_in fields until do_stepwise_t starts ping-ponging
        a.q1_in = (const T*)carve(vec);
        a.pm_in = (const T*)carve(vec);
        a.part_in = (const T*)carve(pg);
    }
    a.C = C;
    a.p = m->p;
    {
        // wide float32 models: the exact-split bf16 matrix-core kernels (lr_wide_bf16.h), 4 or 8 waves per workgroup
        a.wide_bf16 = (m->P > 32 && m->d_xblk) ? wide_engine(m, Cp) : 0;
        a.xblk = static_cast<const uint16_t*>(m->d_xblk);
        a.xblk1 = static_cast<const uint16_t*>(m->d_xblk1);
        a.xblk1h = static_cast<const uint16_t*>(m->d_xblk1h);
        a.xmx = static_cast<const uint16_t*>(m->d_xmx);
    }
    return LR_OK;
}

// lr_eval through the stepwise engine (tall or wide models): load -> partial(value, grad) -> finish
template <typename T, int P>
int do_eval_stepwise_t(lr_model* m, const Plan& pl, hipStream_t st, int64_t C, const void* beta, void* ll, void* lprior,
                       void* lpost, void* grad) {
    if ((C + 63) / 64 > 65535) return fail(LR_ERR_UNSUPPORTED, "the stepwise engine takes at most %lld chains per call (got %lld)", 65535LL * 64, (long long)C);
    lr::TallArgs<T, P> a;
    int rc = setup_tall<T, P>(m, pl, st, C, C, &a);
    if (rc) return rc;
    a.state = static_cast<T*>(const_cast<void*>(beta));
    a.ev_ll = static_cast<T*>(ll);
    a.ev_lprior = static_cast<T*>(lprior);
    a.ev_lpost = static_cast<T*>(lpost);
    a.ev_grad = static_cast<T*>(grad);
    const lr::InstTable* t = m->table;
    rc = t->launch_tall_update(st, lr::KIND_HMC, lr::PH_LOAD, 0, -1, 0, &a);
    if (!rc) rc = t->launch_tall_partial(st, 1, 1, &a);
    if (!rc) rc = t->launch_tall_update(st, lr::KIND_HMC, lr::PH_EVAL, 0, -1, 0, &a);
    if (rc) return fail(LR_ERR_HIP, "stepwise eval launch failed (%d): %s", rc, hipGetErrorString(hipGetLastError()));
    return LR_OK;
}

// wide models: two chain tiles per workgroup of the trajectory kernel beyond this many tiles per CU (one tile per workgroup up to it)
constexpr int kTraj2FromTilesPerCu = 1;

template <typename T, int P>
int do_stepwise_t(lr_model* m, const Plan& pl, hipStream_t st, const RunSpec& rs, const lr_run_opts* o, void* state,
                  double* lp_state, void* out, uint32_t* accepts) {
    const int64_t C = o->n_chains, Cp = plan_count(o);
    // (chain blocks of 16 .. 128 chains are the grid's y / x dimension of the partial kernels: y is a 16-bit quantity)
    if ((C + 63) / 64 > 65535) return fail(LR_ERR_UNSUPPORTED, "the stepwise engine takes at most %lld chains per call (got %lld): split the run into shards (chain_offset)", 65535LL * 64, (long long)C);
    lr::TallArgs<T, P> a;
    int rc = setup_tall<T, P>(m, pl, st, C, Cp, &a);
    if (rc) return rc;
    a.state = static_cast<T*>(state);
    a.lp_state = lp_state;
    a.out = static_cast<T*>(out);
    a.accepts = accepts;
    a.C = C;
    a.chain_offset = o->chain_offset;
    a.seed = o->seed;
    a.p = m->p;
    a.l = rs.l;
    a.step = (T)rs.step;
    for (int j = 0; j < P; ++j) {
        a.a[j] = (T)rs.a[j];
        a.b[j] = (T)rs.b[j];
        a.c[j] = (T)rs.c[j];
    }
    a.stats = lr::StatsArgs{o->stats, o->stats_batch, o->stats_first};
    const lr::InstTable* t = m->table;
    auto U = [&](int phase, int64_t iter, int64_t out_row, int bn) {
        if (!rc) rc = t->launch_tall_update(st, rs.kind, phase, iter, out_row, bn, &a);
    };
    auto K = [&](int v, int g) {
        if (!rc) rc = t->launch_tall_partial(st, v, g, &a);
    };
    // interior leapfrog gradients (HMC): reduced precision where the policy allows and a kernel exists
    const bool bf16_interior = rs.kind == lr::KIND_HMC && o->precision != LR_PREC_FULL &&
                               (m->d_xblk1 != nullptr || m->d_xmx != nullptr);
    const int RS_exact = a.RS, RS_mid = bf16_interior && a.RS_i > 0 ? a.RS_i : a.RS;
    if (!bf16_interior) a.RS_i = 0;
    auto KI = [&]() {
        a.interior = bf16_interior ? 1 : 0;
        LR_STAMPS_ARM(a);
        K(0, 1);
        LR_STAMPS_DISARM(a);
        a.interior = 0;
    };
    // wide models: the whole interior of a trajectory in one launch (k_wide_traj2_bf16): no slice partials, no update launches.  One
    // workgroup streams the whole design per step.  With two chain tiles per workgroup it wins at every chain count from one tile per CU
    // up (us per evaluation, launch per step | trajectory kernel: 12 288 chains 79.6 | 49.6, 16 384: 67.5 | 51.4, 32 768: 120.6 | 101.6,
    // 65 536: 235.9 | 202.7; profiles/r5_cfg5_whole.txt); below that, with one tile per workgroup, by the measured rule further down.
    // LOGREG_DEBUG_OPTS wide_traj=1 / 2 force it with one / two tiles per workgroup, wide_traj=0 forbids it.
    const int64_t traj_tiles = (Cp + 15) / 16;
    // Two chain tiles per workgroup (the same trajectories bit for bit) from the chain count at which one tile per workgroup needs a
    // second round of workgroups.
    // LR_PREC_BF16 (the caller's explicit request for the cheapest interior force): beta in ONE bf16 piece, on the two-tile kernel at any
    // tile count the trajectory path takes -- a third of the MFMAs fewer for ~0.02 of acceptance (lr_wide_bf16.h)
    const bool one_piece = o->precision == LR_PREC_BF16 && m->dbg.wide_traj != 1 && traj_tiles > (int64_t)kTraj2FromTilesPerCu * m->cus;  // (where the two-tile kernel runs anyway; 4096 chains: 16.6 against 16.9 us on the one-tile kernel -- nothing to buy)
    const bool half_ok = m->d_xblk1h != nullptr && m->dbg.wide_f16 != 0;
    // Operand format of every reduced-precision interior kernel of a wide model: rows and beta in ONE f16 piece each where the rows fit f16
    // (11 significant bits against bf16's 8: closer to the exact force than bf16 rows x two bf16 pieces of beta, with a third of the MFMAs
    // fewer -- config 5 whole: 23.9 -> 20.9 us per evaluation, acceptance 0.756 -> 0.758 = the exact interior's), else the bf16 pieces.
    a.traj_fmt = m->dbg.wide_f16 == 2 && half_ok ? 2 : one_piece ? 1 : (half_ok ? 2 : 0);
    a.traj_tiles = (m->dbg.wide_traj == 2 || one_piece) ? 2 : (m->dbg.wide_traj == 1 ? 1 : (traj_tiles > (int64_t)kTraj2FromTilesPerCu * m->cus ? 2 : 1));
    // Below one chain tile per CU the kernel runs with ONE tile per workgroup (k_wide_traj2_bf16<.., 1>; round 5: it replaced round 3's
    // one-tile kernel -- the same trajectories bit for bit, 1.1 - 1.5 x faster).  Its step costs what streaming the image through one
    // CU costs, whatever the chain count; the launch-per-step kernels split the rows over workgroups but pay the launch boundary and the
    // partial hand-over, and degrade from half the chip's tiles up (us per evaluation, trajectory | launch per step, tools/traj_rule_check.py:
    // p=128 n=2000 (500 KB) 1024 chains 5.6 | 7.3; n=3000 (750 KB) 7.3 | 8.0; n=4096 (1 MB) 9.2 | 8.8 at 1024 chains, 10.2 | 12.0 at 2048,
    // 12.1 | 15.8 at 3072; p=64 n=3000 (375 KB) 5.3 | 5.6; n=5000 (625 KB) 7.8 | 6.5 at 1024 chains; n=8000 (1000 KB) 11.6 | 8.0).
    const int64_t image_bytes = (int64_t)m->n * m->P * 2;
    const bool traj = P > 32 && bf16_interior && m->d_xblk1 != nullptr && t->launch_tall_traj != nullptr && rs.l > 1 &&
                      m->dbg.wide_traj != 0 &&
                      (m->dbg.wide_traj >= 1 || traj_tiles >= m->cus ||
                       (traj_tiles >= m->cus / 2 && image_bytes <= (P == 128 ? 1024 : 768) * 1024) ||  // (p = 64, n = 8000 at 2048 chains: 12.6 | 11.5)
                       (traj_tiles >= 3 * m->cus / 4 && image_bytes <= 1024 * 1024) ||
                       image_bytes <= 512 * 1024 || (P == 128 && image_bytes <= 768 * 1024));
    const bool fuse = bf16_interior && a.RS_i > 0 &&
                      ((P > 32 && a.RS_i <= 4) ||                                            // kFuseSlices (lr_wide_bf16.h)
                       (P <= 32 && a.rowsplit_waves == 16 && a.RS_i <= 16 && m->d_xmx));   // kMx16FuseSlices (lr_tall_mx.h)
    T* qb[2] = {a.q1, const_cast<T*>(a.q1_in)};
    T* pb[2] = {a.pm, const_cast<T*>(a.pm_in)};
    T* gb[2] = {a.part_g, const_cast<T*>(a.part_in)};
    int cs = 0, cg = 0;  // which of the pairs holds the current state / the latest partials
    const int kind = rs.kind;
    U(lr::PH_LOAD, 0, -1, 0);
    if (kind == lr::KIND_HMC) K(1, 1);
    else if (kind != lr::KIND_RWMH) K(0, 1);
    U(lr::PH_INIT, o->iter_offset, -1, 0);
    const int64_t total = o->iters * o->thin;
    for (int64_t tt = 0; tt < total && !rc; ++tt) {
        if (kind == lr::KIND_HMC) {
            if (traj) {
                LR_STAMPS_ARM(a);
                if (!rc) rc = t->launch_tall_traj(st, &a);
                LR_STAMPS_DISARM(a);
            } else if (fuse) {
                // row-split interior kernel: every launch but the first finishes the previous leapfrog step in its
                // own prologue (state and partial buffers ping-pong), so the L - 1 interior steps are L - 1
                // launches plus ONE update at the end instead of 2 (L - 1) launches
                for (int i = 0; i < rs.l - 1; ++i) {
                    if (i > 0) {
                        a.q1_in = qb[cs];
                        a.pm_in = pb[cs];
                        a.part_in = gb[cg];
                        cs ^= 1;
                        cg ^= 1;
                        a.q1 = qb[cs];
                        a.pm = pb[cs];
                        a.part_g = gb[cg];
                    }
                    a.fuse_mid = i > 0;
                    KI();
                    a.fuse_mid = 0;
                }
                if (rs.l > 1) {
                    a.RS = RS_mid;
                    a.part_f32 = sizeof(T) == 8;  // (float64 models: the interior kernel's partials are float32)
                    U(lr::PH_MID, 0, -1, 0);
                    a.part_f32 = 0;
                    a.RS = RS_exact;
                }
            } else {
                for (int i = 0; i < rs.l - 1; ++i) {
                    KI();
                    a.RS = RS_mid;  // the update sums as many slice partials as the partial kernel just wrote
                    a.part_f32 = sizeof(T) == 8 && bf16_interior;  // (float64 models: those partials are float32)
                    U(lr::PH_MID, 0, -1, 0);
                    a.part_f32 = 0;
                    a.RS = RS_exact;
                }
            }
            K(1, 1);
        } else if (kind == lr::KIND_MALA) {
            K(1, 1);
        } else if (kind == lr::KIND_RWMH) {
            K(1, 0);
        } else {
            K(0, 1);
        }
        const int64_t out_row = ((tt + 1) % o->thin == 0) ? (tt + 1) / o->thin - 1 : -1;
        U(lr::PH_END, o->iter_offset + tt, out_row, tt + 1 < total);
    }
    U(lr::PH_STORE, 0, -1, 0);
    if (rc) return fail(LR_ERR_HIP, "stepwise launch failed (%d): %s", rc, hipGetErrorString(hipGetLastError()));
    return LR_OK;
}

#define LR_DISPATCH_TP(m, FN, ...)                                                       \
    do {                                                                                 \
        if ((m)->dtype == LR_F32) {                                                      \
            switch ((m)->P) {                                                            \
            case 4: return FN<float, 4>(__VA_ARGS__);                                    \
            case 8: return FN<float, 8>(__VA_ARGS__);                                    \
            case 16: return FN<float, 16>(__VA_ARGS__);                                  \
            case 32: return FN<float, 32>(__VA_ARGS__);                                  \
            }                                                                            \
        } else {                                                                         \
            switch ((m)->P) {                                                            \
            case 4: return FN<double, 4>(__VA_ARGS__);                                   \
            case 8: return FN<double, 8>(__VA_ARGS__);                                   \
            case 16: return FN<double, 16>(__VA_ARGS__);                                 \
            case 32: return FN<double, 32>(__VA_ARGS__);                                 \
            }                                                                            \
        }                                                                                \
        return fail(LR_ERR_UNSUPPORTED, "unsupported padded width %d", (m)->P);          \
    } while (0)

int do_eval_stepwise(lr_model* m, const Plan& pl, hipStream_t st, int64_t C, const void* beta, void* ll, void* lprior,
                     void* lpost, void* grad);

int do_eval(lr_model* m, const Plan& pl, hipStream_t st, int64_t C, const void* beta, void* ll, void* lprior,
            void* lpost, void* grad) {
    if (pl.mode == lr::MODE_STEPWISE) return do_eval_stepwise(m, pl, st, C, beta, ll, lprior, lpost, grad);
    LR_DISPATCH_TP(m, do_eval_t, m, pl, st, C, beta, ll, lprior, lpost, grad);
}

#define LR_DISPATCH_STEP(m, FN, ...)                                                     \
    do {                                                                                 \
        if ((m)->dtype == LR_F32 && (m)->P == 64) return FN<float, 64>(__VA_ARGS__);     \
        if ((m)->dtype == LR_F32 && (m)->P == 128) return FN<float, 128>(__VA_ARGS__);   \
        if ((m)->dtype == LR_F64 && (m)->P == 64) return FN<double, 64>(__VA_ARGS__);    \
        if ((m)->dtype == LR_F64 && (m)->P == 128) return FN<double, 128>(__VA_ARGS__);  \
        LR_DISPATCH_TP(m, FN, __VA_ARGS__);                                              \
    } while (0)

int do_stepwise(lr_model* m, const Plan& pl, hipStream_t st, const RunSpec& rs, const lr_run_opts* o, void* state,
                double* lp_state, void* out, uint32_t* accepts) {
    LR_DISPATCH_STEP(m, do_stepwise_t, m, pl, st, rs, o, state, lp_state, out, accepts);
}

int do_eval_stepwise(lr_model* m, const Plan& pl, hipStream_t st, int64_t C, const void* beta, void* ll, void* lprior,
                     void* lpost, void* grad) {
    LR_DISPATCH_STEP(m, do_eval_stepwise_t, m, pl, st, C, beta, ll, lprior, lpost, grad);
}

int do_chain(lr_model* m, const Plan& pl, hipStream_t st, const RunSpec& rs, const lr_run_opts* o, void* state,
             double* lp_state, void* out, uint32_t* accepts) {
    if (pl.mode == lr::MODE_STEPWISE) return do_stepwise(m, pl, st, rs, o, state, lp_state, out, accepts);
    LR_DISPATCH_TP(m, do_chain_t, m, pl, st, rs, o, state, lp_state, out, accepts);
}

}  // namespace
