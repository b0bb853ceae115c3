// engine_harness.cpp -- TEST INFRASTRUCTURE: the host engine of liblogreg_hip.so (lr_api.hip, lr_engine.h, lr_model.h, lr_plan.h)
// driven through its own C ABI on the stub HIP runtime (tests/host/hip_stub.cpp), built with -fsanitize=address,undefined or
// -fsanitize=thread by tests/test_engine_sanitizers.py.  Kernels do not run (a launch is a validated no-op), so nothing here checks a
// number: it checks that every path of the host engine -- model images of every kind, one- and two-part plans with their fork / join
// events, the stepwise engines and their workspaces, two chain sets on two streams, statistics, the Hessian, every error return, a
// failing allocation at every point of model creation -- touches only memory it owns, frees what it allocates, and waits only on
// events it recorded.
//   engine_harness all        every scenario on one thread
//   engine_harness threads    two host threads, one model / stream / chain set each, running concurrently (ThreadSanitizer)
#include "logreg_hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
long hipstub_launches();
long hipstub_launches_on(void* stream);
long hipstub_launches_of(const char* name_part);
long hipstub_bad_waits();
long hipstub_bad_launches();
long hipstub_live_allocs();
long hipstub_live_streams();
long hipstub_live_events();
long hipstub_mallocs();
void hipstub_fail_malloc_at(long nth);
void hipstub_set_devices(int n);
long hipstub_work_on_device(int d);
long hipstub_wrong_device();
}

static int g_fail = 0;
static long g_deliberate_bad_waits = 0;  // (scenario_errors asks for the elapsed time of an event it never recorded)
#define EXPECT(cond, ...)                                                              \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::printf("FAIL %s:%d: %s  -- ", __FILE__, __LINE__, #cond);             \
            std::printf(__VA_ARGS__);                                                  \
            std::printf("  (last error: %s)\n", lr_last_error());                      \
            ++g_fail;                                                                  \
        }                                                                              \
    } while (0)

struct Data {
    std::vector<double> X, y, sd;
    int64_t n;
    int p;
};
static Data make_data(int64_t n, int p, unsigned seed) {
    Data d;
    d.n = n; d.p = p;
    d.X.resize((size_t)n * p); d.y.resize(n); d.sd.assign(p, 2.0);
    unsigned long long s = seed * 2654435761ull + 12345;
    auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
    for (int64_t i = 0; i < n; ++i) {
        d.X[(size_t)i * p] = 1.0;
        for (int j = 1; j < p; ++j) d.X[(size_t)i * p + j] = 2.0 * u() - 1.0;
        d.y[i] = u() < 0.5 ? 0.0 : 1.0;
    }
    return d;
}
struct Dev {  // a device buffer through the ABI's own allocator
    void* p = nullptr;
    explicit Dev(size_t bytes) { if (lr_malloc(0, bytes, &p) != LR_OK) p = nullptr; }
    ~Dev() { if (p) lr_free(0, p); }
};

// every kernel family on one model, host buffers and device buffers, with and without statistics
static void run_all_kinds(lr_model* m, const Data& d, int64_t C, int dtype, void* stream, int precision, int mode = LR_MODE_AUTO, int group = 0) {
    const size_t es = dtype == LR_F32 ? 4 : 8;
    std::vector<unsigned char> state((size_t)C * d.p * es, 0), out((size_t)2 * C * d.p * es);
    std::vector<double> lp(C, -INFINITY), vec(d.p, 1.0);
    std::vector<uint32_t> acc(C, 0);
    lr_run_opts o{};
    o.n_chains = C; o.thin = 2; o.iters = 2; o.seed = 7; o.mode = mode; o.group = group; o.precision = precision; o.stream = nullptr;
    EXPECT(lr_run_hmc(m, state.data(), 0.01, 3, vec.data(), &o, out.data(), acc.data()) == LR_OK, "hmc host C=%lld", (long long)C);
    EXPECT(lr_run_mala(m, state.data(), lp.data(), 1e-3, vec.data(), &o, out.data(), acc.data()) == LR_OK, "mala host");
    EXPECT(lr_run_rwmh(m, state.data(), lp.data(), vec.data(), &o, out.data(), acc.data()) == LR_OK, "rwmh host");
    EXPECT(lr_run_ul(m, state.data(), 1e-3, vec.data(), &o, nullptr, acc.data()) == LR_OK, "ul host, no samples kept");
    // on-device buffers on the caller's stream, with a statistics window
    Dev dstate((size_t)C * d.p * es), dlp((size_t)C * 8), dout((size_t)2 * C * d.p * es), dacc((size_t)C * 4), dstats((size_t)2 * C * 2 * d.p * 8);
    EXPECT(dstate.p && dlp.p && dout.p && dacc.p && dstats.p, "device buffers");
    o.on_device = 1; o.stream = stream; o.stats = (double*)dstats.p; o.stats_batch = 1; o.stats_first = 0; o.stats_slots = 2;
    EXPECT(lr_run_hmc(m, dstate.p, 0.01, 4, vec.data(), &o, dout.p, (uint32_t*)dacc.p) == LR_OK, "hmc device + stats");
    o.iter_offset = 4; o.stats_first = 0;
    EXPECT(lr_run_mala(m, dstate.p, (double*)dlp.p, 1e-3, vec.data(), &o, nullptr, (uint32_t*)dacc.p) == LR_OK, "mala device, stats only");
    std::vector<double> piv(d.p, 0.0), sums((size_t)LR_STATS_ROWS * d.p);
    EXPECT(lr_stats_reduce(0, (double*)dstats.p, C, d.p, 1, 2, piv.data(), sums.data(), stream) == LR_OK, "stats reduce");
    EXPECT(lr_stream_sync(0, stream) == LR_OK, "sync");
    // closures
    std::vector<unsigned char> beta((size_t)C * d.p * es, 0), ll((size_t)C * es), grad((size_t)C * d.p * es);
    lr_run_opts e{};
    e.n_chains = C; e.mode = LR_MODE_AUTO;
    EXPECT(lr_eval(m, beta.data(), ll.data(), ll.data(), ll.data(), grad.data(), &e) == LR_OK, "eval");
    EXPECT(lr_eval(m, beta.data(), nullptr, nullptr, ll.data(), nullptr, &e) == LR_OK, "eval lpost only");
}

static void scenario_models() {
    struct Shape { int64_t n; int p; int64_t C; const char* what; };
    const Shape shapes[] = {
        {200, 8, 64, "Pima-sized: rows in registers"}, {200, 8, 4096, "one wave per SIMD"}, {200, 8, 5120, "a two-part plan (head + remainder on a side stream)"},
        {200, 8, 20480, "matrix-core head + remainder"}, {1500, 8, 4096, "rows in LDS / matrix-core operands in LDS"}, {3000, 8, 2048, "operand images in device memory"},
        {20000, 8, 256, "tall: stepwise engine + matrix-pipe interior image"}, {300, 24, 600, "17 <= p <= 32"}, {1000, 12, 4096, "9 <= p <= 16 matrix-core"},
        {600, 64, 300, "wide p = 64: row-split / trajectory kernels"}, {700, 128, 1100, "wide p = 128, several chain blocks"}, {37, 3, 5, "tiny"}, {1, 1, 1, "n = p = C = 1"}};
    void* stream = nullptr;
    EXPECT(lr_stream_create(0, &stream) == LR_OK, "stream");
    for (const Shape& s : shapes) {
        const Data d = make_data(s.n, s.p, (unsigned)(s.n + s.p));
        for (int dtype : {LR_F32, LR_F64}) {
            lr_model* m = nullptr;
            EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), dtype, 0, &m) == LR_OK, "create %s dtype %d", s.what, dtype);
            if (!m) continue;
            int64_t n; int32_t p, dt, dev, pp, fmt;
            EXPECT(lr_model_info(m, &n, &p, &dt, &dev, &pp) == LR_OK && n == d.n && p == d.p && dt == dtype && pp >= p, "info");
            EXPECT(lr_model_interior_format(m, &fmt) == LR_OK, "interior format");
            char buf[64];
            EXPECT(lr_model_debug_opts(m, buf, sizeof buf) == LR_OK, "debug opts");
            for (int prec : {LR_PREC_AUTO, LR_PREC_FULL}) run_all_kinds(m, d, s.C, dtype, stream, prec);
            // the plan as the caller sees it, for a run and for a shard of it
            lr_run_opts o{};
            o.n_chains = s.C; o.thin = 1; o.iters = 1; o.mode = LR_MODE_AUTO;
            lr_plan_info pi{};
            EXPECT(lr_plan_run_info(m, LR_KIND_HMC, &o, &pi) == LR_OK, "plan info");
            if (s.C >= 4) {  // a shard planned as the whole run is (plan_chains / plan_first), straddling a two-part split if there is one
                lr_run_opts sh = o;
                sh.n_chains = s.C / 2; sh.chain_offset = 1000 + s.C / 4; sh.plan_chains = (int32_t)s.C; sh.plan_first = 1000; sh.thin = 2; sh.iters = 1; sh.seed = 3;
                const size_t es = dtype == LR_F32 ? 4 : 8;
                std::vector<unsigned char> st((size_t)sh.n_chains * d.p * es, 0), out((size_t)sh.n_chains * d.p * es);
                std::vector<double> vec(d.p, 1.0);
                std::vector<uint32_t> acc(sh.n_chains, 0);
                EXPECT(lr_run_hmc(m, st.data(), 0.01, 2, vec.data(), &sh, out.data(), acc.data()) == LR_OK, "planned shard of %s", s.what);
            }
            if (d.p <= 32 && d.n >= 8) {
                std::vector<double> b(d.p, 0.0), g(d.p), h((size_t)d.p * d.p);
                double lpost = 0;
                EXPECT(lr_hessian(m, b.data(), &lpost, g.data(), h.data(), stream) == LR_OK, "hessian");
            }
            lr_model_destroy(m);
        }
    }
    EXPECT(lr_stream_destroy(0, stream) == LR_OK, "stream destroy");
}

// forced engines: every (mode, group) the planner accepts on a small model, both dtypes
static void scenario_forced_variants() {
    const Data d = make_data(200, 8, 5);
    for (int dtype : {LR_F32, LR_F64}) {
        lr_model* m = nullptr;
        EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), dtype, 0, &m) == LR_OK, "create");
        int ok = 0;
        for (int mode : {LR_MODE_REG, LR_MODE_LDS, LR_MODE_GLOBAL, LR_MODE_MFMA, LR_MODE_STEPWISE, LR_MODE_MIXED})
            for (int g : {0, 1, 2, 4, 8, 16, 32, 64}) {
                int32_t mo, go, ro;
                if (lr_plan(m, 300, g, mode, &mo, &go, &ro) != LR_OK) continue;  // (an LR_ERR_* with a message: not a finding)
                ++ok;
                run_all_kinds(m, d, 300, dtype, nullptr, LR_PREC_AUTO, mode, g);
            }
        EXPECT(ok >= 8, "forced variants accepted: %d", ok);
        lr_model_destroy(m);
    }
}

// two chain sets of ONE model on two streams, interleaved launches (per-stream workspaces and side slots), then a second model
static void scenario_two_streams() {
    const Data d = make_data(200, 8, 11), w = make_data(600, 64, 12);
    lr_model *m = nullptr, *mw = nullptr;
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), LR_F32, 0, &m) == LR_OK, "create");
    EXPECT(lr_model_create(w.X.data(), w.y.data(), w.n, w.p, w.sd.data(), LR_F32, 0, &mw) == LR_OK, "create wide");
    void *s1 = nullptr, *s2 = nullptr;
    EXPECT(lr_stream_create(0, &s1) == LR_OK && lr_stream_create(0, &s2) == LR_OK, "streams");
    const int64_t C = 5120;  // (a two-part plan: each launch forks to a side stream and joins)
    Dev st1((size_t)C * 8 * 4), st2((size_t)C * 8 * 4), acc((size_t)C * 4), wst((size_t)300 * 64 * 4), wacc(300 * 4);
    std::vector<double> vec(8, 1.0), wvec(64, 1.0);
    lr_run_opts o{};
    o.n_chains = C; o.thin = 1; o.iters = 1; o.seed = 1; o.on_device = 1; o.mode = LR_MODE_AUTO; o.precision = LR_PREC_FULL;
    lr_run_opts ow = o;
    ow.n_chains = 300; ow.precision = LR_PREC_AUTO;
    const long before1 = hipstub_launches_on(s1), before2 = hipstub_launches_on(s2);
    for (int it = 0; it < 4; ++it) {
        o.iter_offset = it;
        o.stream = s1;
        EXPECT(lr_run_hmc(m, st1.p, 0.01, 3, vec.data(), &o, nullptr, (uint32_t*)acc.p) == LR_OK, "set 1");
        o.stream = s2;
        EXPECT(lr_run_hmc(m, st2.p, 0.01, 3, vec.data(), &o, nullptr, (uint32_t*)acc.p) == LR_OK, "set 2");
        ow.stream = it & 1 ? s1 : s2;  // the stepwise engine's workspaces are per stream as well
        ow.iter_offset = it;
        EXPECT(lr_run_hmc(mw, wst.p, 0.01, 4, wvec.data(), &ow, nullptr, (uint32_t*)wacc.p) == LR_OK, "wide on alternating streams");
    }
    EXPECT(hipstub_launches_on(s1) > before1 && hipstub_launches_on(s2) > before2, "launches went to the callers' streams");
    EXPECT(lr_stream_sync(0, s1) == LR_OK && lr_stream_sync(0, s2) == LR_OK, "sync");
    lr_model_destroy(mw);
    lr_model_destroy(m);
    EXPECT(lr_stream_destroy(0, s1) == LR_OK && lr_stream_destroy(0, s2) == LR_OK, "destroy streams");
}

static void scenario_errors() {
    const Data d = make_data(50, 4, 3);
    lr_model* m = nullptr;
    std::vector<double> bad = d.y;
    bad[3] = 0.5;
    EXPECT(lr_model_create(nullptr, d.y.data(), d.n, d.p, d.sd.data(), LR_F32, 0, &m) == LR_ERR_INVALID, "NULL X");
    EXPECT(lr_model_create(d.X.data(), bad.data(), d.n, d.p, d.sd.data(), LR_F32, 0, &m) == LR_ERR_INVALID, "y not 0/1");
    EXPECT(lr_model_create(d.X.data(), d.y.data(), 0, d.p, d.sd.data(), LR_F32, 0, &m) == LR_ERR_INVALID, "n = 0");
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, 1000, d.sd.data(), LR_F32, 0, &m) == LR_ERR_UNSUPPORTED, "p too large");
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), 7, 0, &m) == LR_ERR_INVALID, "dtype");
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), LR_F32, 5, &m) == LR_ERR_HIP, "device ordinal");
    std::vector<double> sd0 = d.sd;
    sd0[1] = 0.0;
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, sd0.data(), LR_F32, 0, &m) == LR_ERR_INVALID, "prior sd 0");
    std::vector<double> Xn = d.X;
    Xn[7] = NAN;
    EXPECT(lr_model_create(Xn.data(), d.y.data(), d.n, d.p, d.sd.data(), LR_F32, 0, &m) == LR_ERR_INVALID, "NaN in X");
    EXPECT(m == nullptr, "no model came out of a failed create");
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), LR_F32, 0, &m) == LR_OK, "create");
    std::vector<float> st(64 * 4, 0.f), out(64 * 4);
    std::vector<double> lp(64, 0.0), vec(4, 1.0), neg(4, -1.0);
    std::vector<uint32_t> acc(64, 0);
    lr_run_opts o{};
    o.n_chains = 64; o.thin = 1; o.iters = 1; o.mode = LR_MODE_AUTO;
    EXPECT(lr_run_hmc(nullptr, st.data(), 0.1, 2, vec.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "NULL model");
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), nullptr, out.data(), acc.data()) == LR_ERR_INVALID, "NULL opts");
    EXPECT(lr_run_hmc(m, nullptr, 0.1, 2, vec.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "NULL state");
    EXPECT(lr_run_hmc(m, st.data(), -0.1, 2, vec.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "eps < 0");
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 0, vec.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "l = 0");
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, neg.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "dmm < 0");
    EXPECT(lr_run_mala(m, st.data(), nullptr, 1e-3, vec.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "mala without lp_state");
    EXPECT(lr_run_mala(m, st.data(), lp.data(), 0.0, vec.data(), &o, out.data(), acc.data()) == LR_ERR_INVALID, "dt = 0");
    EXPECT(lr_run_rwmh(m, st.data(), lp.data(), nullptr, &o, out.data(), acc.data()) == LR_ERR_INVALID, "NULL prop_sd");
    lr_run_opts b = o;
    b.n_chains = 0;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "0 chains");
    b = o; b.thin = 0;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "thin = 0");
    b = o; b.precision = 9;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "precision");
    b = o; b.group = 3;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "group 3");
    b = o; b.mode = LR_MODE_MFMA;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_UNSUPPORTED, "no matrix-core kernel at p = 4");
    b = o; b.chain_offset = (int64_t)1 << 33;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "chain id beyond 32 bits");
    // the containment rule of a planned shard: [chain_offset, chain_offset + n) inside [plan_first, plan_first + plan_chains)
    b = o; b.plan_chains = 100; b.plan_first = 50; b.chain_offset = 40;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "shard starts before the planned run");
    b.chain_offset = 100;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "shard ends beyond the planned run");
    b.chain_offset = 60; b.n_chains = 40;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_OK, "a shard inside the planned run");
    b = o; b.plan_chains = -1;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "plan_chains < 0");
    std::vector<double> stats(2 * 64 * 2 * 4);
    b = o; b.stats = stats.data(); b.stats_batch = 0; b.stats_slots = 2;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "stats_batch = 0");
    b.stats_batch = 1; b.stats_first = 2; b.iters = 1;
    EXPECT(lr_run_hmc(m, st.data(), 0.1, 2, vec.data(), &b, out.data(), acc.data()) == LR_ERR_INVALID, "statistics window overrun");
    lr_run_opts e{};
    e.n_chains = 64;
    EXPECT(lr_eval(m, nullptr, out.data(), nullptr, nullptr, nullptr, &e) == LR_ERR_INVALID, "eval NULL beta");
    int32_t a, g2, r;
    EXPECT(lr_plan(m, 0, 0, LR_MODE_AUTO, &a, &g2, &r) == LR_ERR_INVALID, "plan for 0 chains");
    EXPECT(lr_plan_run(m, 9, &o, &a, &g2, &r) == LR_ERR_INVALID, "plan for an unknown kind");
    EXPECT(lr_stats_reduce(0, nullptr, 64, 4, 1, 1, vec.data(), vec.data(), nullptr) == LR_ERR_INVALID, "stats reduce NULL");
    float ms;
    void* ev = nullptr;
    EXPECT(lr_event_create(0, &ev) == LR_OK, "event");
    const long bw = hipstub_bad_waits();
    EXPECT(lr_event_elapsed_ms(0, ev, ev, &ms) != LR_OK, "elapsed time of an event never recorded");
    g_deliberate_bad_waits += hipstub_bad_waits() - bw;
    EXPECT(lr_event_record(0, ev, nullptr) == LR_OK && lr_event_elapsed_ms(0, ev, ev, &ms) == LR_OK, "record + elapsed");
    EXPECT(lr_event_destroy(0, ev) == LR_OK, "event destroy");
    EXPECT(std::strlen(lr_last_error()) > 0, "the last failure left a message");
    lr_model_destroy(m);
    lr_model_destroy(nullptr);  // (as free(NULL))
    hipstub_set_devices(0);
    EXPECT(lr_device_count() == 0, "no device");
    EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), LR_F32, 0, &m) == LR_ERR_HIP, "create without a device");
    hipstub_set_devices(1);
}

// the k-th device allocation of lr_model_create fails, for every k the creation makes: LR_ERR_NOMEM, nothing leaked, no handle returned
static void scenario_failing_allocations() {
    struct Shape { int64_t n; int p; int dtype; };
    for (const Shape& s : {Shape{200, 8, LR_F32}, Shape{3000, 8, LR_F32}, Shape{20000, 8, LR_F32}, Shape{700, 128, LR_F32}, Shape{700, 128, LR_F64}, Shape{300, 24, LR_F64}}) {
        const Data d = make_data(s.n, s.p, 99);
        lr_model* m = nullptr;
        const long m0 = hipstub_mallocs();
        EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), s.dtype, 0, &m) == LR_OK, "create");
        const long made = hipstub_mallocs() - m0;
        lr_model_destroy(m);
        const long live0 = hipstub_live_allocs();
        for (long k = 1; k <= made; ++k) {
            m = nullptr;
            hipstub_fail_malloc_at(k);
            const int rc = lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), s.dtype, 0, &m);
            hipstub_fail_malloc_at(-1);
            EXPECT(rc == LR_ERR_NOMEM && m == nullptr, "allocation %ld of %ld fails (n=%lld p=%d): rc %d", k, made, (long long)s.n, s.p, rc);
            EXPECT(hipstub_live_allocs() == live0, "allocation %ld of %ld fails: %ld device buffers leaked", k, made, hipstub_live_allocs() - live0);
            if (m) lr_model_destroy(m);
        }
        // ... and a failing allocation inside a run (sample buffer, statistics, workspaces): an error, then the model still works
        EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), s.dtype, 0, &m) == LR_OK, "create again");
        const size_t es = s.dtype == LR_F32 ? 4 : 8;
        const int64_t C = 128;
        std::vector<unsigned char> st((size_t)C * d.p * es, 0), out((size_t)C * d.p * es);
        std::vector<double> vec(d.p, 1.0);
        std::vector<uint32_t> acc(C, 0);
        lr_run_opts o{};
        o.n_chains = C; o.thin = 1; o.iters = 1; o.mode = LR_MODE_AUTO;
        const long r0 = hipstub_mallocs();
        EXPECT(lr_run_hmc(m, st.data(), 0.01, 2, vec.data(), &o, out.data(), acc.data()) == LR_OK, "run");
        const long in_run = hipstub_mallocs() - r0;
        lr_model_destroy(m);
        for (long k = 1; k <= in_run; ++k) {
            EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), s.dtype, 0, &m) == LR_OK, "fresh model");
            const long before = hipstub_live_allocs();
            hipstub_fail_malloc_at(k);
            const int rc = lr_run_hmc(m, st.data(), 0.01, 2, vec.data(), &o, out.data(), acc.data());
            hipstub_fail_malloc_at(-1);
            EXPECT(rc == LR_ERR_NOMEM || rc == LR_ERR_HIP, "allocation %ld of %ld inside the run fails: rc %d", k, in_run, rc);
            EXPECT(lr_run_hmc(m, st.data(), 0.01, 2, vec.data(), &o, out.data(), acc.data()) == LR_OK, "the model runs after a failed run");
            lr_model_destroy(m);
            EXPECT(hipstub_live_allocs() <= before - 1, "a failed run leaked past the model's destruction");
        }
    }
}

// A rank of a multi-GPU job works on device LOCAL_RANK, not 0 (bench.py, logreg_amd.distributed): everything the library allocates,
// creates and launches for a model on device 1 must happen with device 1 current -- a path that forgot its hipSetDevice would put a
// workspace or a launch on device 0 and only an 8-GPU node would ever show it.
static void scenario_second_device() {
    hipstub_set_devices(2);
    EXPECT(lr_device_count() == 2 && lr_device_cus(1) == 256, "two devices");
    char info[256];
    EXPECT(lr_device_info(1, info, sizeof info) == LR_OK && std::strstr(info, "pci=") != nullptr, "device info of device 1");
    void* stream = nullptr;
    const long zero0 = hipstub_work_on_device(0);
    EXPECT(lr_stream_create(1, &stream) == LR_OK, "stream on device 1");
    struct Shape { int64_t n; int p; int64_t C; };
    for (const Shape& s : {Shape{200, 8, 5120}, Shape{20000, 8, 256}, Shape{600, 64, 300}, Shape{300, 24, 600}}) {
        const Data d = make_data(s.n, s.p, 77);
        for (int dtype : {LR_F32, LR_F64}) {
            lr_model* m = nullptr;
            EXPECT(lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), dtype, 1, &m) == LR_OK, "create on device 1");
            if (!m) continue;
            int64_t n; int32_t p, dt, dev, pp;
            EXPECT(lr_model_info(m, &n, &p, &dt, &dev, &pp) == LR_OK && dev == 1, "the model reports device 1");
            const size_t es = dtype == LR_F32 ? 4 : 8;
            void *st = nullptr, *lp = nullptr, *out = nullptr, *acc = nullptr, *stats = nullptr;
            EXPECT(lr_malloc(1, (size_t)s.C * d.p * es, &st) == LR_OK && lr_malloc(1, (size_t)s.C * 8, &lp) == LR_OK && lr_malloc(1, (size_t)s.C * d.p * es, &out) == LR_OK &&
                   lr_malloc(1, (size_t)s.C * 4, &acc) == LR_OK && lr_malloc(1, (size_t)s.C * 2 * d.p * 8, &stats) == LR_OK, "buffers on device 1");
            std::vector<double> vec(d.p, 1.0);
            lr_run_opts o{};
            o.n_chains = s.C; o.thin = 1; o.iters = 1; o.on_device = 1; o.stream = stream; o.mode = LR_MODE_AUTO;
            o.stats = (double*)stats; o.stats_batch = 1; o.stats_slots = 1;
            for (int prec : {LR_PREC_AUTO, LR_PREC_FULL}) {
                o.precision = prec;
                EXPECT(lr_run_hmc(m, st, 0.01, 3, vec.data(), &o, out, (uint32_t*)acc) == LR_OK, "hmc on device 1");
                EXPECT(lr_run_mala(m, st, (double*)lp, 1e-3, vec.data(), &o, out, (uint32_t*)acc) == LR_OK, "mala on device 1");
            }
            std::vector<double> piv(d.p, 0.0), sums((size_t)LR_STATS_ROWS * d.p);
            EXPECT(lr_stats_reduce(1, (double*)stats, s.C, d.p, 1, 1, piv.data(), sums.data(), stream) == LR_OK, "stats reduce on device 1");
            std::vector<unsigned char> host((size_t)s.C * d.p * es);
            EXPECT(lr_memcpy_d2h(1, host.data(), out, host.size(), stream) == LR_OK && lr_memset(1, acc, 0, (size_t)s.C * 4, stream) == LR_OK, "copies on device 1");
            // (host-pointer convenience path: the library stages through its own device buffers)
            std::vector<unsigned char> hs((size_t)64 * d.p * es, 0), ho((size_t)64 * d.p * es);
            std::vector<uint32_t> ha(64, 0);
            lr_run_opts h{};
            h.n_chains = 64; h.thin = 1; h.iters = 1; h.mode = LR_MODE_AUTO;
            EXPECT(lr_run_hmc(m, hs.data(), 0.01, 2, vec.data(), &h, ho.data(), ha.data()) == LR_OK, "host buffers, model on device 1");
            if (d.p <= 32) {
                std::vector<double> b(d.p, 0.0), g(d.p), hh((size_t)d.p * d.p);
                double lpost;
                EXPECT(lr_hessian(m, b.data(), &lpost, g.data(), hh.data(), stream) == LR_OK, "hessian on device 1");
            }
            EXPECT(lr_stream_sync(1, stream) == LR_OK, "sync");
            for (void* q : {st, lp, out, acc, stats}) EXPECT(lr_free(1, q) == LR_OK, "free");
            lr_model_destroy(m);
        }
    }
    EXPECT(lr_stream_destroy(1, stream) == LR_OK, "stream destroy");
    EXPECT(hipstub_work_on_device(0) == zero0, "%ld allocations / creations / launches happened with device 0 current while working on device 1",
           hipstub_work_on_device(0) - zero0);
    EXPECT(hipstub_work_on_device(1) > 200, "work counted on device 1: %ld", hipstub_work_on_device(1));
    hipstub_set_devices(1);
}

static void thread_body(int id, int* fails) {
    const Data d = id == 0 ? make_data(200, 8, 21) : make_data(600, 64, 22);
    lr_model* m = nullptr;
    void* stream = nullptr;
    int bad = 0;
    bad += lr_model_create(d.X.data(), d.y.data(), d.n, d.p, d.sd.data(), id == 0 ? LR_F32 : LR_F64, 0, &m) != LR_OK;
    bad += lr_stream_create(0, &stream) != LR_OK;
    const int64_t C = id == 0 ? 5120 : 300;
    const size_t es = id == 0 ? 4 : 8;
    Dev st((size_t)C * d.p * es), acc((size_t)C * 4), out((size_t)C * d.p * es);
    std::vector<double> vec(d.p, 1.0);
    lr_run_opts o{};
    o.n_chains = C; o.thin = 1; o.iters = 1; o.on_device = 1; o.stream = stream; o.mode = LR_MODE_AUTO; o.precision = id == 0 ? LR_PREC_FULL : LR_PREC_AUTO;
    for (int it = 0; it < 20 && m; ++it) {
        o.iter_offset = it;
        bad += lr_run_hmc(m, st.p, 0.01, 3, vec.data(), &o, out.p, (uint32_t*)acc.p) != LR_OK;
        if (it % 5 == 0) bad += lr_stream_sync(0, stream) != LR_OK;
    }
    lr_model_destroy(m);
    bad += lr_stream_destroy(0, stream) != LR_OK;
    *fails = bad;
}

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    if (what == "threads") {
        int f0 = 0, f1 = 0;
        std::thread a(thread_body, 0, &f0), b(thread_body, 1, &f1);
        a.join();
        b.join();
        EXPECT(f0 == 0 && f1 == 0, "thread failures %d %d", f0, f1);
    } else {
        scenario_models();
        scenario_forced_variants();
        scenario_two_streams();
        scenario_errors();
        scenario_failing_allocations();
        scenario_second_device();
    }
    EXPECT(hipstub_wrong_device() == 0, "%ld uses of another device's stream / event", hipstub_wrong_device());
    EXPECT(hipstub_bad_waits() == g_deliberate_bad_waits, "%ld waits on events that were never recorded", hipstub_bad_waits() - g_deliberate_bad_waits);
    EXPECT(hipstub_bad_launches() == 0, "%ld launches with an invalid configuration or a dead stream", hipstub_bad_launches());
    EXPECT(hipstub_live_allocs() == 0, "%ld device buffers alive at exit", hipstub_live_allocs());
    EXPECT(hipstub_live_streams() == 0 && hipstub_live_events() == 0, "%ld streams, %ld events alive at exit", hipstub_live_streams(), hipstub_live_events());
    std::printf("engine harness (%s): %ld kernel launches (%ld k_chain*, %ld k_tall*, %ld k_wide*), %ld device allocations, %d failures\n", what.c_str(), hipstub_launches(),
                hipstub_launches_of("k_chain"), hipstub_launches_of("k_tall"), hipstub_launches_of("k_wide"), hipstub_mallocs(), g_fail);
    return g_fail ? 1 : 0;
}
