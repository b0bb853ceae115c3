// lr_api.hip -- the entry points of the C ABI declared in include/logreg_hip.h: argument checking, model creation (device copies
// and operand images), the run / eval / plan calls and the device-memory helpers.  Host-side only.  The layers below it:
//   lr_model.h   the model handle, error reporting, variant tables          lr_plan.h    which kernel variant runs a request (data)
//   lr_engine.h  kernel argument packing, workspaces, the stepwise driver   lr_inst*.hip the launches, one unit per (dtype, width)
// All arithmetic of the path runs in the kernels (lr_kernels.h, lr_mfma.h, lr_tall*.h, lr_wide*.h).
#include "../../include/logreg_hip.h"

#include <hip/hip_runtime.h>
#define LR_STAMPS_HOST  // this unit also gets the host side of the development instrumentation (lr_stamps.h: empty in production builds)

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lr_inst.h"
#include "lr_kernels.h"
#include "lr_hessian.h"
#include "lr_mfma.h"
#include "lr_stats.h"
#include "lr_tall.h"
#include "lr_tall_mx.h"
#include "lr_wide_bf16.h"

#include "lr_model.h"

#include "lr_plan.h"
#include "lr_engine.h"

namespace {

int check_opts(const lr_model* m, const lr_run_opts* o, bool run) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!o) return fail(LR_ERR_INVALID, "opts is NULL");
    if (o->n_chains <= 0) return fail(LR_ERR_INVALID, "n_chains must be positive (got %lld)", (long long)o->n_chains);
    if (o->plan_chains < 0) return fail(LR_ERR_INVALID, "plan_chains must be 0 (= n_chains) or positive (got %d)", o->plan_chains);
    if (o->plan_chains > 0 && run && (o->plan_first < 0 || o->plan_first > o->chain_offset || o->chain_offset + o->n_chains > o->plan_first + o->plan_chains))
        return fail(LR_ERR_INVALID, "chains [%lld, %lld) are not inside the planned run [%lld, %lld) (plan_first, plan_chains)", (long long)o->chain_offset,
                    (long long)(o->chain_offset + o->n_chains), (long long)o->plan_first, (long long)(o->plan_first + o->plan_chains));
    if (const int rcg = check_group_for(m, o->group, o->mode)) return rcg;
    if (run) {
        if (o->thin <= 0 || o->iters < 0) return fail(LR_ERR_INVALID, "thin must be > 0 and iters >= 0");
        if (o->chain_offset < 0 || o->iter_offset < 0) return fail(LR_ERR_INVALID, "offsets must be >= 0");
        if ((uint64_t)(o->chain_offset + o->n_chains) > 0xFFFFFFFFull)
            return fail(LR_ERR_INVALID, "global chain ids must fit 32 bits");
        if (o->precision < LR_PREC_AUTO || o->precision > LR_PREC_BF16)
            return fail(LR_ERR_INVALID, "precision must be LR_PREC_AUTO/FULL/BF16");
        if (o->stats) {
            if (o->stats_batch < 1 || o->stats_first < 0 || o->stats_slots < 1)
                return fail(LR_ERR_INVALID, "stats needs stats_batch >= 1, stats_first >= 0, stats_slots >= 1");
            if (o->stats_first + o->iters > o->stats_slots * o->stats_batch)
                return fail(LR_ERR_INVALID, "stats buffer too small: kept samples %lld..%lld need more than %lld slots of %lld",
                            (long long)o->stats_first, (long long)(o->stats_first + o->iters), (long long)o->stats_slots,
                            (long long)o->stats_batch);
        }
    }
    return LR_OK;
}

// Host-pointer convenience path: stage through device buffers, run synchronously, copy back.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 1;
        return hipMalloc(&p, bytes) == hipSuccess ? 0 : -1;
    }
};

int run_common(lr_model* m, const RunSpec& rs, const lr_run_opts* o, void* state, double* lp_state, void* out,
               uint32_t* accepts) {
    int rc = check_opts(m, o, true);
    if (rc) return rc;
    if (!state) return fail(LR_ERR_INVALID, "state is NULL");
    const bool threaded = rs.kind == lr::KIND_RWMH || rs.kind == lr::KIND_MALA;
    if (threaded && !lp_state) return fail(LR_ERR_INVALID, "lp_state is required for RWMH/MALA");
    LR_HIP(hipSetDevice(m->device));
    Plan pl;
    rc = make_plan(m, plan_count(o), o->group, o->mode, &pl, false, rs.kind == lr::KIND_HMC && o->precision != LR_PREC_FULL, rs.kind,
                   o->precision == LR_PREC_AUTO);
    if (rc) return rc;
    if (o->iters == 0) return LR_OK;
    if (o->on_device) return do_chain(m, pl, (hipStream_t)o->stream, rs, o, state, threaded ? lp_state : nullptr, out, accepts);

    const size_t es = m->esize();
    const size_t sbytes = (size_t)o->n_chains * m->p * es;
    const size_t obytes = out ? (size_t)o->iters * o->n_chains * m->p * es : 0;
    DevBuf ds, dl, dout, dacc, dstats;
    if (ds.alloc(sbytes) || dl.alloc(o->n_chains * sizeof(double)) || dout.alloc(obytes) ||
        dacc.alloc(o->n_chains * sizeof(uint32_t)))
        return fail(LR_ERR_NOMEM, "device allocation failed (%zu bytes of samples)", obytes);
    lr_run_opts od = *o;  // device-side view of the options: the statistics buffer is staged like the other arrays
    const size_t stbytes = o->stats ? (size_t)o->stats_slots * o->n_chains * 2 * m->p * sizeof(double) : 0;
    if (o->stats) {
        if (dstats.alloc(stbytes)) return fail(LR_ERR_NOMEM, "device allocation failed (%zu bytes of statistics)", stbytes);
        LR_HIP(hipMemcpy(dstats.p, o->stats, stbytes, hipMemcpyHostToDevice));
        od.stats = static_cast<double*>(dstats.p);
    }
    LR_HIP(hipMemcpy(ds.p, state, sbytes, hipMemcpyHostToDevice));
    if (threaded) LR_HIP(hipMemcpy(dl.p, lp_state, o->n_chains * sizeof(double), hipMemcpyHostToDevice));
    if (accepts) LR_HIP(hipMemcpy(dacc.p, accepts, o->n_chains * sizeof(uint32_t), hipMemcpyHostToDevice));
    rc = do_chain(m, pl, nullptr, rs, &od, ds.p, threaded ? (double*)dl.p : nullptr, out ? dout.p : nullptr,
                  accepts ? (uint32_t*)dacc.p : nullptr);
    if (rc) return rc;
    LR_HIP(hipDeviceSynchronize());
    if (o->stats) LR_HIP(hipMemcpy(o->stats, dstats.p, stbytes, hipMemcpyDeviceToHost));
    LR_HIP(hipMemcpy(state, ds.p, sbytes, hipMemcpyDeviceToHost));
    if (threaded) LR_HIP(hipMemcpy(lp_state, dl.p, o->n_chains * sizeof(double), hipMemcpyDeviceToHost));
    if (out) LR_HIP(hipMemcpy(out, dout.p, obytes, hipMemcpyDeviceToHost));
    if (accepts) LR_HIP(hipMemcpy(accepts, dacc.p, o->n_chains * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return LR_OK;
}

int positive_vec(const char* name, const double* v, int p) {
    if (!v) return fail(LR_ERR_INVALID, "%s is NULL", name);
    for (int j = 0; j < p; ++j)
        if (!(v[j] > 0) || !std::isfinite(v[j])) return fail(LR_ERR_INVALID, "%s[%d] must be finite and > 0", name, j);
    return LR_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

const char* lr_last_error(void) { return g_err; }

#ifndef LR_BUILD_ID
#define LR_BUILD_ID "unversioned"
#endif
const char* lr_build_id(void) { return LR_BUILD_ID; }
int lr_sizeof_run_opts(void) { return (int)sizeof(lr_run_opts); }


int lr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lr_device_cus(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(LR_ERR_HIP, "hipGetDeviceProperties failed");
    return prop.multiProcessorCount;
}

int lr_device_info(int device, char* buf, int len) {
    if (!buf || len <= 0) return fail(LR_ERR_INVALID, "NULL / empty buffer");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(LR_ERR_HIP, "hipGetDeviceProperties failed");
    char pci[32] = "?";
    if (hipDeviceGetPCIBusId(pci, sizeof pci, device) != hipSuccess) snprintf(pci, sizeof pci, "?");
    char uuid[33];
    for (int i = 0; i < 16; ++i) snprintf(uuid + 2 * i, 3, "%02x", (unsigned)(unsigned char)prop.uuid.bytes[i]);
    snprintf(buf, (size_t)len, "pci=%s uuid=%s name=%s cus=%d", pci, uuid, prop.name, prop.multiProcessorCount);
    return LR_OK;
}

int lr_model_create(const double* X, const double* y, int64_t n, int32_t p, const double* prior_sd, int32_t dtype,
                    int32_t device, lr_model** out) {
    if (!X || !y || !prior_sd || !out) return fail(LR_ERR_INVALID, "NULL argument");
    if (n <= 0 || p <= 0) return fail(LR_ERR_INVALID, "n and p must be positive");
    if (p > kMaxP) return fail(LR_ERR_UNSUPPORTED, "p=%d > %d is not supported", p, kMaxP);
    if (dtype != LR_F32 && dtype != LR_F64) return fail(LR_ERR_INVALID, "dtype must be LR_F32 or LR_F64");
    int rc = positive_vec("prior_sd", prior_sd, p);
    if (rc) return rc;
    for (int64_t i = 0; i < n; ++i)
        if (y[i] != 0.0 && y[i] != 1.0) return fail(LR_ERR_INVALID, "y[%lld]=%g is not 0/1", (long long)i, y[i]);
    int ndev = 0;
    LR_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(LR_ERR_HIP, "device %d not available (%d visible)", device, ndev);
    LR_HIP(hipSetDevice(device));

    lr_model* m = new lr_model();
    {
        char bad[64];
        if (!parse_debug_opts(std::getenv("LOGREG_DEBUG_OPTS"), &m->dbg, bad, sizeof bad)) {
            delete m;
            return fail(LR_ERR_INVALID, "LOGREG_DEBUG_OPTS: unknown or out-of-range item '%s' (keys: residency_cap=0|1, tall_mx16=0|1, "
                                        "wide_traj=0|1|2, wide_waves=4|8, wide_f16=0|1|2)", bad);
        }
    }
    m->device = device;
    m->dtype = dtype;
    m->n = n;
    m->p = p;
    m->P = padded_width(p);
    const ModelImages images = model_images(n, m->P, dtype);
    m->table = find_table(dtype, m->P);
    if (!m->table) { delete m; return fail(LR_ERR_UNSUPPORTED, "no kernels for dtype=%d padded p=%d", dtype, m->P); }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) m->cus = prop.multiProcessorCount;
    m->lprior_const = 0;
    for (int j = 0; j < kMaxP; ++j) m->inv_var[j] = 0;
    for (int j = 0; j < p; ++j) {
        m->inv_var[j] = 1.0 / (prior_sd[j] * prior_sd[j]);
        m->lprior_const += -std::log(prior_sd[j]) - 0.91893853320467274178;
    }
    // signed rows  xs_i = (2 y_i - 1) x_i, zero-padded to P columns, in the compute dtype
    const size_t elems = (size_t)n * m->P;
    std::vector<unsigned char> host(elems * m->esize());
    for (int64_t i = 0; i < n; ++i) {
        const double s = 2.0 * y[i] - 1.0;
        for (int j = 0; j < m->P; ++j) {
            const double v = j < p ? s * X[i * p + j] : 0.0;
            if (!std::isfinite(v)) { delete m; return fail(LR_ERR_INVALID, "X[%lld,%d] is not finite", (long long)i, j); }
            if (dtype == LR_F32) reinterpret_cast<float*>(host.data())[i * m->P + j] = (float)v;
            else reinterpret_cast<double*>(host.data())[i * m->P + j] = v;
        }
    }
    if (hipMalloc(&m->d_rows, host.size()) != hipSuccess) { delete m; return fail(LR_ERR_NOMEM, "hipMalloc rows failed"); }
    if (hipMemcpy(m->d_rows, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(m->d_rows);
        delete m;
        return fail(LR_ERR_HIP, "hipMemcpy rows failed");
    }
    if (m->P <= 32 && dtype == LR_F32) {
        // the rows once more as twisted row pairs, for the kernels that take them through the scalar unit:
        // pair k, coordinates (j, j+1), j even:  [k][j] = (A_j, B_{j+1}),  [k][j+1] = (A_{j+1}, B_j)  with
        // A = row 2k, B = row 2k+1 (a zero row closes an odd n)
        const float* hrows = reinterpret_cast<const float*>(host.data());
        const int64_t npair = (n + 1) / 2;
        const int PP = m->P;
        std::vector<float> tw((size_t)npair * PP * 2, 0.0f);
        auto at = [&](int64_t r, int j) { return r < n ? hrows[r * PP + j] : 0.0f; };
        for (int64_t k = 0; k < npair; ++k)
            for (int j = 0; j < PP; j += 2) {
                float* q = tw.data() + ((size_t)k * PP + j) * 2;
                q[0] = at(2 * k, j);
                q[1] = at(2 * k + 1, j + 1);
                q[2] = at(2 * k, j + 1);
                q[3] = at(2 * k + 1, j);
            }
        if (hipMalloc(&m->d_rows_tw, tw.size() * 4) != hipSuccess ||
            hipMemcpy(m->d_rows_tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            lr_model_destroy(m);
            return fail(LR_ERR_NOMEM, "allocating the row-pair image (%zu bytes) failed", tw.size() * 4);
        }
    }
    // narrow models the planner would ever send to the stepwise engine by itself (rows beyond 64 KB): two-piece bf16 tile images
    // for the interior HMC steps on the matrix pipe.  Smaller models run the engine only when forced (mode = STEPWISE), then in
    // fp32 throughout, and carry no image.
    if (images.tall_mx) {
        std::vector<float> rounded;
        if (dtype != LR_F32) {
            rounded.resize(elems);
            for (size_t i = 0; i < elems; ++i) rounded[i] = (float)reinterpret_cast<const double*>(host.data())[i];
        }
        const float* hrows = dtype == LR_F32 ? reinterpret_cast<const float*>(host.data()) : rounded.data();
        const int64_t ntile = (n + 31) / 32 * 2;
        std::vector<uint16_t> img((size_t)ntile * (m->P / 8) * lr::kMxSetElems);
        if (m->P == 8) lr::tall_mx_prepare<8>(hrows, n, img.data());
        else if (m->P == 16) lr::tall_mx_prepare<16>(hrows, n, img.data());
        else lr::tall_mx_prepare<32>(hrows, n, img.data());
        if (hipMalloc(&m->d_xmx, img.size() * 2) != hipSuccess ||
            hipMemcpy(m->d_xmx, img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
            lr_model_destroy(m);
            return fail(LR_ERR_NOMEM, "allocating the bf16 tile images (%zu bytes) failed", img.size() * 2);
        }
    }
    if (images.mf_end) {  // (up to 8192 rows: profiles/r2_midn_lds_mfma.txt)
        // the matrix-core chain kernel would keep its bf16 operands in LDS: fp32 operand images for its end points
        const float* hrows = reinterpret_cast<const float*>(host.data());
        const size_t fl = (size_t)((n + 15) / 16) * 64 *
                          (m->P == 8 ? lr::mf_image_floats<8>() : (m->P == 16 ? lr::mf_image_floats<16>() : lr::mf_image_floats<32>()));
        std::vector<float> img(fl);
        if (m->P == 8) lr::mf_image_prepare<8>(hrows, n, img.data());
        else if (m->P == 16) lr::mf_image_prepare<16>(hrows, n, img.data());
        else lr::mf_image_prepare<32>(hrows, n, img.data());
        if (hipMalloc(&m->d_xmf, fl * 4) != hipSuccess || hipMemcpy(m->d_xmf, img.data(), fl * 4, hipMemcpyHostToDevice) != hipSuccess) {
            lr_model_destroy(m);
            return fail(LR_ERR_NOMEM, "allocating the fp32 operand images (%zu bytes) failed", fl * 4);
        }
        if (model_wants_xms(m) && m->table->mfma_image_bytes && m->table->launch_mfma_image) {
            const size_t ib = m->table->mfma_image_bytes(n);  // beyond LDS: the interior operands, built on the device once
            if (hipMalloc(&m->d_xms, ib) != hipSuccess || m->table->launch_mfma_image(nullptr, m->d_rows, n, m->d_xms) != 0 ||
                hipDeviceSynchronize() != hipSuccess) {
                lr_model_destroy(m);
                return fail(LR_ERR_NOMEM, "building the bf16 operand images (%zu bytes) failed", ib);
            }
        }
    }
    if (images.wide) {  // wide float32 models: bf16-piece block images for the exact-split matrix-core kernels
        const float* hrows = reinterpret_cast<const float*>(host.data());
        const int64_t nblk = (n + 31) / 32;
        const size_t elems_blk = m->P == 64 ? (size_t)lr::WideBf16Geom<64>::BUF : (size_t)lr::WideBf16Geom<128>::BUF;
        std::vector<uint16_t> img((size_t)nblk * elems_blk);
        if (m->P == 64) lr::wide_bf16_prepare<64>(hrows, n, img.data());
        else lr::wide_bf16_prepare<128>(hrows, n, img.data());
        if (hipMalloc(&m->d_xblk, img.size() * 2) != hipSuccess ||
            hipMemcpy(m->d_xblk, img.data(), img.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
            lr_model_destroy(m);
            return fail(LR_ERR_NOMEM, "allocating the bf16 block images (%zu bytes) failed", img.size() * 2);
        }
    }
    if (images.wide1) {  // ... and the one-piece image (float64 models: from the rows rounded to float32)
        std::vector<float> rounded;
        if (dtype != LR_F32) {
            rounded.resize(elems);
            for (size_t i = 0; i < elems; ++i) rounded[i] = (float)reinterpret_cast<const double*>(host.data())[i];
        }
        const float* hrows = dtype == LR_F32 ? reinterpret_cast<const float*>(host.data()) : rounded.data();
        const int64_t nblk = (n + 31) / 32;
        const size_t elems_blk1 = m->P == 64 ? (size_t)lr::WideBf16Geom<64>::BUF1 : (size_t)lr::WideBf16Geom<128>::BUF1;
        std::vector<uint16_t> img1((size_t)nblk * elems_blk1);
        if (m->P == 64) lr::wide_bf16_prepare_rne<64>(hrows, n, img1.data());
        else lr::wide_bf16_prepare_rne<128>(hrows, n, img1.data());
        if (hipMalloc(&m->d_xblk1, img1.size() * 2) != hipSuccess ||
            hipMemcpy(m->d_xblk1, img1.data(), img1.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
            lr_model_destroy(m);
            return fail(LR_ERR_NOMEM, "allocating the single-piece bf16 block images (%zu bytes) failed", img1.size() * 2);
        }
        {  // the interior kernels' half-precision image, where the rows fit its range (lr_wide_bf16.h)
            const bool fits = m->P == 64 ? lr::wide_f16_prepare_rne<64>(hrows, n, img1.data()) : lr::wide_f16_prepare_rne<128>(hrows, n, img1.data());
            if (fits && (hipMalloc(&m->d_xblk1h, img1.size() * 2) != hipSuccess ||
                         hipMemcpy(m->d_xblk1h, img1.data(), img1.size() * 2, hipMemcpyHostToDevice) != hipSuccess)) {
                lr_model_destroy(m);
                return fail(LR_ERR_NOMEM, "allocating the single-piece f16 block images (%zu bytes) failed", img1.size() * 2);
            }
        }
    }
    *out = m;
    return LR_OK;
}

void lr_model_destroy(lr_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->d_rows) (void)hipFree(m->d_rows);
    if (m->d_rows_tw) (void)hipFree(m->d_rows_tw);
    for (auto& e : m->ws)
        if (e.p) (void)hipFree(e.p);
    if (m->d_xblk) (void)hipFree(m->d_xblk);
    if (m->d_xblk1) (void)hipFree(m->d_xblk1);
    if (m->d_xblk1h) (void)hipFree(m->d_xblk1h);
    if (m->d_xmx) (void)hipFree(m->d_xmx);
    if (m->d_xmf) (void)hipFree(m->d_xmf);
    if (m->d_xms) (void)hipFree(m->d_xms);
    for (auto& e : m->sides) {
        (void)hipStreamDestroy(e.stream);
        (void)hipEventDestroy(e.ev_fork);
        (void)hipEventDestroy(e.ev_join);
    }
    delete m;
}

int lr_model_debug_opts(const lr_model* m, char* buf, int len) {
    if (!m || !buf || len <= 0) return fail(LR_ERR_INVALID, "NULL argument / empty buffer");
    const DebugOpts& d = m->dbg;
    if (d.is_default()) snprintf(buf, (size_t)len, "%s", "");
    else snprintf(buf, (size_t)len, "residency_cap=%d,tall_mx16=%d,wide_traj=%d,wide_waves=%d,wide_f16=%d", d.residency_cap, d.tall_mx16, d.wide_traj, d.wide_waves, d.wide_f16);
    return LR_OK;
}

int lr_model_interior_format(const lr_model* m, int32_t* format) {
    if (!m || !format) return fail(LR_ERR_INVALID, "NULL argument");
    if (m->P > 32) *format = (m->d_xblk1h && m->dbg.wide_f16 != 0) ? LR_INTERIOR_F16 : (m->d_xblk1 ? LR_INTERIOR_BF16 : LR_INTERIOR_NONE);
    else *format = m->P >= 8 ? LR_INTERIOR_BF16 : LR_INTERIOR_NONE;  // (the matrix-core kernels of lr_mfma.h / lr_mfma_f64.h / lr_tall_mx.h)
    return LR_OK;
}

int lr_model_info(const lr_model* m, int64_t* n, int32_t* p, int32_t* dtype, int32_t* device, int32_t* padded_p) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (n) *n = m->n;
    if (p) *p = m->p;
    if (dtype) *dtype = m->dtype;
    if (device) *device = m->device;
    if (padded_p) *padded_p = m->P;
    return LR_OK;
}

int lr_plan(const lr_model* m, int64_t n_chains, int32_t group, int32_t mode, int32_t* mode_out, int32_t* group_out,
            int32_t* rows_out) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (n_chains <= 0) return fail(LR_ERR_INVALID, "n_chains must be positive (got %lld)", (long long)n_chains);
    if (const int rcg = check_group_for(m, group, mode)) return rcg;
    Plan pl;
    const int rc = make_plan(m, n_chains, group, mode, &pl);
    if (rc) return rc;
    if (mode_out) *mode_out = pl.mode;
    if (group_out) *group_out = pl.G;
    if (rows_out) *rows_out = pl.R;
    return LR_OK;
}

int lr_plan_run_info(const lr_model* m, int32_t kind, const lr_run_opts* o, lr_plan_info* out) {
    if (!out) return fail(LR_ERR_INVALID, "out is NULL");
    int rc = check_opts(m, o, false);
    if (rc) return rc;
    if (kind < LR_KIND_RWMH || kind > LR_KIND_UL) return fail(LR_ERR_INVALID, "kind must be one of LR_KIND_*");
    Plan pl;
    rc = make_plan(m, plan_count(o), o->group, o->mode, &pl, false, kind == LR_KIND_HMC && o->precision != LR_PREC_FULL, kind,
                   o->precision == LR_PREC_AUTO);
    if (rc) return rc;
    *out = lr_plan_info{pl.mode, pl.G, pl.R, pl.G2, pl.R2, pl.split, pl.mode2, 0};
    return LR_OK;
}

int lr_plan_run(const lr_model* m, int32_t kind, const lr_run_opts* o, int32_t* mode_out, int32_t* group_out, int32_t* rows_out) {
    int rc = check_opts(m, o, false);
    if (rc) return rc;
    if (kind < LR_KIND_RWMH || kind > LR_KIND_UL) return fail(LR_ERR_INVALID, "kind must be one of LR_KIND_*");
    Plan pl;
    rc = make_plan(m, plan_count(o), o->group, o->mode, &pl, false, kind == LR_KIND_HMC && o->precision != LR_PREC_FULL, kind,
                   o->precision == LR_PREC_AUTO);
    if (rc) return rc;
    if (mode_out) *mode_out = pl.mode;
    if (group_out) *group_out = pl.G;
    if (rows_out) *rows_out = pl.R;
    return LR_OK;
}

int lr_eval(lr_model* m, const void* beta, void* ll, void* lprior, void* lpost, void* grad, const lr_run_opts* o) {
    int rc = check_opts(m, o, false);
    if (rc) return rc;
    if (!beta) return fail(LR_ERR_INVALID, "beta is NULL");
    LR_HIP(hipSetDevice(m->device));
    Plan pl;
    rc = make_plan(m, o->n_chains, o->group, o->mode, &pl, true);
    if (rc) return rc;
    const int64_t C = o->n_chains;
    if (o->on_device) return do_eval(m, pl, (hipStream_t)o->stream, C, beta, ll, lprior, lpost, grad);
    const size_t es = m->esize();
    DevBuf db, dll, dlpr, dlpo, dg;
    if (db.alloc(C * m->p * es) || dll.alloc(C * es) || dlpr.alloc(C * es) || dlpo.alloc(C * es) || dg.alloc(C * m->p * es))
        return fail(LR_ERR_NOMEM, "device allocation failed");
    LR_HIP(hipMemcpy(db.p, beta, C * m->p * es, hipMemcpyHostToDevice));
    rc = do_eval(m, pl, nullptr, C, db.p, ll ? dll.p : nullptr, lprior ? dlpr.p : nullptr, lpost ? dlpo.p : nullptr,
                 grad ? dg.p : nullptr);
    if (rc) return rc;
    LR_HIP(hipDeviceSynchronize());
    if (ll) LR_HIP(hipMemcpy(ll, dll.p, C * es, hipMemcpyDeviceToHost));
    if (lprior) LR_HIP(hipMemcpy(lprior, dlpr.p, C * es, hipMemcpyDeviceToHost));
    if (lpost) LR_HIP(hipMemcpy(lpost, dlpo.p, C * es, hipMemcpyDeviceToHost));
    if (grad) LR_HIP(hipMemcpy(grad, dg.p, C * m->p * es, hipMemcpyDeviceToHost));
    return LR_OK;
}

int lr_run_rwmh(lr_model* m, void* state, double* lp_state, const double* prop_sd, const lr_run_opts* o, void* out,
                uint32_t* accepts) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!prop_sd) return fail(LR_ERR_INVALID, "prop_sd is NULL");
    RunSpec rs{};
    rs.kind = lr::KIND_RWMH;
    for (int j = 0; j < m->p; ++j) {
        if (!(prop_sd[j] >= 0) || !std::isfinite(prop_sd[j])) return fail(LR_ERR_INVALID, "prop_sd[%d] must be finite and >= 0", j);
        rs.a[j] = prop_sd[j];
    }
    return run_common(m, rs, o, state, lp_state, out, accepts);
}

static int langevin_spec(lr_model* m, int kind, double dt, const double* pre, RunSpec* rs) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!(dt > 0) || !std::isfinite(dt)) return fail(LR_ERR_INVALID, "dt must be finite and > 0");
    const int rc = positive_vec("pre", pre, m->p);
    if (rc) return rc;
    rs->kind = kind;
    rs->step = dt;
    for (int j = 0; j < m->p; ++j) {
        rs->a[j] = 0.5 * pre[j] * dt;
        rs->b[j] = std::sqrt(pre[j]) * std::sqrt(dt);
        rs->c[j] = 1.0 / (rs->b[j] * rs->b[j]);
    }
    return LR_OK;
}

int lr_run_mala(lr_model* m, void* state, double* lp_state, double dt, const double* pre, const lr_run_opts* o, void* out,
                uint32_t* accepts) {
    RunSpec rs{};
    const int rc = langevin_spec(m, lr::KIND_MALA, dt, pre, &rs);
    if (rc) return rc;
    return run_common(m, rs, o, state, lp_state, out, accepts);
}

int lr_run_ul(lr_model* m, void* state, double dt, const double* pre, const lr_run_opts* o, void* out, uint32_t* accepts) {
    RunSpec rs{};
    const int rc = langevin_spec(m, lr::KIND_UL, dt, pre, &rs);
    if (rc) return rc;
    return run_common(m, rs, o, state, nullptr, out, accepts);
}

int lr_run_hmc(lr_model* m, void* state, double eps, int32_t l, const double* dmm, const lr_run_opts* o, void* out,
               uint32_t* accepts) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!(eps > 0) || !std::isfinite(eps)) return fail(LR_ERR_INVALID, "eps must be finite and > 0");
    if (l < 1) return fail(LR_ERR_INVALID, "l must be >= 1");
    const int rc = positive_vec("dmm", dmm, m->p);
    if (rc) return rc;
    RunSpec rs{};
    rs.kind = lr::KIND_HMC;
    rs.step = eps;
    rs.l = l;
    for (int j = 0; j < m->p; ++j) {
        rs.a[j] = std::sqrt(dmm[j]);
        rs.b[j] = eps / dmm[j];
        rs.c[j] = 1.0 / dmm[j];
    }
    return run_common(m, rs, o, state, nullptr, out, accepts);
}

int lr_hessian(lr_model* m, const double* beta, double* lpost, double* grad, double* hess, void* stream) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!beta) return fail(LR_ERR_INVALID, "beta is NULL");
    LR_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const int p = m->p, NE = p * (p + 1) / 2, width = NE + p + 1;
    const int64_t nblocks = (m->n + lr::kHessRows - 1) / lr::kHessRows;
    DevBuf scratch;
    if (scratch.alloc(((size_t)nblocks * width + width + p) * sizeof(double))) return fail(LR_ERR_NOMEM, "lr_hessian scratch");
    double* d_part = static_cast<double*>(scratch.p);
    double* d_sums = d_part + (size_t)nblocks * width;
    double* d_beta = d_sums + width;
    LR_HIP(hipMemcpyAsync(d_beta, beta, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
    const size_t lds = ((size_t)lr::kHessRows * p + 3 * lr::kHessRows + p) * sizeof(double);
    if (m->dtype == LR_F32)
        hipLaunchKernelGGL((lr::k_hess_partial<float>), dim3((unsigned)nblocks), dim3(256), lds, st,
                           static_cast<const float*>(m->d_rows), m->n, m->P, p, d_beta, d_part);
    else
        hipLaunchKernelGGL((lr::k_hess_partial<double>), dim3((unsigned)nblocks), dim3(256), lds, st,
                           static_cast<const double*>(m->d_rows), m->n, m->P, p, d_beta, d_part);
    LR_HIP(hipGetLastError());
    hipLaunchKernelGGL(lr::k_hess_final, dim3((width + 255) / 256), dim3(256), 0, st, d_part, nblocks, width, d_sums);
    LR_HIP(hipGetLastError());
    std::vector<double> h(width);
    LR_HIP(hipMemcpyAsync(h.data(), d_sums, (size_t)width * sizeof(double), hipMemcpyDeviceToHost, st));
    LR_HIP(hipStreamSynchronize(st));
    if (hess) {
        int e = 0;
        for (int a = 0; a < p; ++a)
            for (int c = a; c < p; ++c, ++e) hess[a * p + c] = hess[c * p + a] = h[e];
        for (int j = 0; j < p; ++j) hess[j * p + j] += m->inv_var[j];
    }
    if (grad)
        for (int j = 0; j < p; ++j) grad[j] = h[NE + j] - beta[j] * m->inv_var[j];
    if (lpost) {
        double quad = 0.0;
        for (int j = 0; j < p; ++j) quad += beta[j] * beta[j] * m->inv_var[j];
        *lpost = h[NE + p] + m->lprior_const - 0.5 * quad;
    }
    return LR_OK;
}

int lr_stats_reduce(int device, const double* stats, int64_t n_chains, int32_t p, int64_t batch, int64_t kept,
                    const double* pivot, double* sums, void* stream) {
    if (!stats || !pivot || !sums) return fail(LR_ERR_INVALID, "NULL argument");
    if (n_chains <= 0 || p <= 0 || p > kMaxP || batch < 1 || kept < 0)
        return fail(LR_ERR_INVALID, "lr_stats_reduce: need n_chains > 0, 0 < p <= %d, batch >= 1, kept >= 0", kMaxP);
    LR_HIP(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    int PW = 1;
    while (PW < p) PW *= 2;
    const int CY = 256 / PW;
    const int64_t nblocks = (n_chains + CY - 1) / CY;
    const size_t part_bytes = (size_t)nblocks * lr::kStatsRows * p * sizeof(double);
    const size_t tail_bytes = (size_t)(lr::kStatsRows + 1) * p * sizeof(double);  // final sums + the pivot
    DevBuf scratch;
    if (scratch.alloc(part_bytes + tail_bytes)) return fail(LR_ERR_NOMEM, "lr_stats_reduce scratch");
    double* d_part = static_cast<double*>(scratch.p);
    double* d_sums = d_part + (size_t)nblocks * lr::kStatsRows * p;
    double* d_piv = d_sums + (size_t)lr::kStatsRows * p;
    LR_HIP(hipMemcpyAsync(d_piv, pivot, (size_t)p * sizeof(double), hipMemcpyHostToDevice, st));
    const dim3 grid((unsigned)nblocks), block(PW, CY);
    switch (PW) {
#define LR_STATS_CASE(W) \
    case W: hipLaunchKernelGGL((lr::k_stats_partial<W>), grid, block, 0, st, stats, n_chains, (int)p, batch, kept, d_piv, d_part); break;
        LR_STATS_CASE(1) LR_STATS_CASE(2) LR_STATS_CASE(4) LR_STATS_CASE(8) LR_STATS_CASE(16) LR_STATS_CASE(32)
        LR_STATS_CASE(64) LR_STATS_CASE(128)
#undef LR_STATS_CASE
    }
    LR_HIP(hipGetLastError());
    hipLaunchKernelGGL(lr::k_stats_final, dim3((lr::kStatsRows * p + 255) / 256), dim3(256), 0, st, d_part, nblocks, (int)p, d_sums);
    LR_HIP(hipGetLastError());
    LR_HIP(hipMemcpyAsync(sums, d_sums, (size_t)lr::kStatsRows * p * sizeof(double), hipMemcpyDeviceToHost, st));
    LR_HIP(hipStreamSynchronize(st));
    return LR_OK;
}

// ---- device memory / stream / event helpers -------------------------------------------------------
int lr_malloc(int device, uint64_t bytes, void** dptr) {
    if (!dptr) return fail(LR_ERR_INVALID, "dptr is NULL");
    LR_HIP(hipSetDevice(device));
    if (hipMalloc(dptr, bytes ? bytes : 1) != hipSuccess) return fail(LR_ERR_NOMEM, "hipMalloc(%llu) failed", (unsigned long long)bytes);
    return LR_OK;
}
int lr_free(int device, void* dptr) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipFree(dptr));
    return LR_OK;
}
int lr_memcpy_h2d(int device, void* dst, const void* src, uint64_t bytes, void* stream) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    LR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return LR_OK;
}
int lr_memcpy_d2h(int device, void* dst, const void* src, uint64_t bytes, void* stream) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    LR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return LR_OK;
}
int lr_memset(int device, void* dst, int value, uint64_t bytes, void* stream) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return LR_OK;
}
int lr_stream_create(int device, void** stream) {
    if (!stream) return fail(LR_ERR_INVALID, "stream is NULL");
    LR_HIP(hipSetDevice(device));
    hipStream_t s;
    LR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return LR_OK;
}
int lr_stream_destroy(int device, void* stream) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipStreamDestroy((hipStream_t)stream));
    return LR_OK;
}
int lr_stream_sync(int device, void* stream) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipStreamSynchronize((hipStream_t)stream));
    return LR_OK;
}
int lr_event_create(int device, void** event) {
    if (!event) return fail(LR_ERR_INVALID, "event is NULL");
    LR_HIP(hipSetDevice(device));
    hipEvent_t e;
    LR_HIP(hipEventCreate(&e));
    *event = e;
    return LR_OK;
}
int lr_event_destroy(int device, void* event) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipEventDestroy((hipEvent_t)event));
    return LR_OK;
}
int lr_event_record(int device, void* event, void* stream) {
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return LR_OK;
}
int lr_event_elapsed_ms(int device, void* start, void* stop, float* ms) {
    if (!ms) return fail(LR_ERR_INVALID, "ms is NULL");
    LR_HIP(hipSetDevice(device));
    LR_HIP(hipEventSynchronize((hipEvent_t)stop));
    LR_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return LR_OK;
}

// ---- the C-level exchange: RCCL, resolved at first use (no link-time dependency; see include/logreg_hip.h)
}  // extern "C"
#include <dlfcn.h>
// (RCCL's header only where the ROCm installation has it -- the library is resolved with dlopen at first use either way; without the
//  header, the handful of declarations this file needs, with the ABI values of rccl.h)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint8 = 1, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif
struct lr_comm {
    ncclComm_t comm;
    int rank, world, device;
};
namespace {
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
Rccl g_rccl;
int rccl_load() {
    if (g_rccl.h) return LR_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail(LR_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
    Rccl r;
    r.h = h;
#define LR_SYM(field, name)                                                     \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));              \
    if (!r.field) return fail(LR_ERR_UNSUPPORTED, "RCCL symbol %s not found", name);
    LR_SYM(get_unique_id, "ncclGetUniqueId")
    LR_SYM(comm_init_rank, "ncclCommInitRank")
    LR_SYM(comm_destroy, "ncclCommDestroy")
    LR_SYM(group_start, "ncclGroupStart")
    LR_SYM(group_end, "ncclGroupEnd")
    LR_SYM(send, "ncclSend")
    LR_SYM(recv, "ncclRecv")
    LR_SYM(all_reduce, "ncclAllReduce")
    LR_SYM(error_string, "ncclGetErrorString")
#undef LR_SYM
    g_rccl = r;
    return LR_OK;
}
#define LR_NCCL(call)                                                                                       \
    do {                                                                                                    \
        const ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess) return fail(LR_ERR_HIP, "%s failed: %s", #call, g_rccl.error_string(r_));    \
    } while (0)
}  // namespace
extern "C" {

int lr_comm_unique_id(void* id) {
    static_assert(LR_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "identifier size");
    if (!id) return fail(LR_ERR_INVALID, "id is NULL");
    if (int rc = rccl_load()) return rc;
    ncclUniqueId u;
    LR_NCCL(g_rccl.get_unique_id(&u));
    std::memcpy(id, u.internal, LR_COMM_ID_BYTES);
    return LR_OK;
}
int lr_comm_create(const void* id, int32_t rank, int32_t world, int device, lr_comm** out) {
    if (!id || !out) return fail(LR_ERR_INVALID, "id / out is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail(LR_ERR_INVALID, "rank %d of world %d", rank, world);
    if (int rc = rccl_load()) return rc;
    LR_HIP(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, LR_COMM_ID_BYTES);
    ncclComm_t c;
    LR_NCCL(g_rccl.comm_init_rank(&c, world, u, rank));
    *out = new lr_comm{c, rank, world, device};
    return LR_OK;
}
int lr_comm_destroy(lr_comm* comm) {
    if (!comm) return LR_OK;
    if (g_rccl.h) (void)g_rccl.comm_destroy(comm->comm);
    delete comm;
    return LR_OK;
}
int lr_gather(lr_comm* comm, const void* send, void* recv, uint64_t bytes, int32_t root, void* stream) {
    if (!comm || !send) return fail(LR_ERR_INVALID, "comm / send is NULL");
    if (root < 0 || root >= comm->world) return fail(LR_ERR_INVALID, "root %d of world %d", root, comm->world);
    if (comm->rank == root && !recv) return fail(LR_ERR_INVALID, "recv is NULL on the root rank");
    LR_HIP(hipSetDevice(comm->device));
    // every rank sends its block to the root, the root receives world blocks in rank order: ONE group (xGMI point-to-point)
    LR_NCCL(g_rccl.group_start());
    LR_NCCL(g_rccl.send(send, bytes, ncclUint8, root, comm->comm, (hipStream_t)stream));
    if (comm->rank == root)
        for (int r = 0; r < comm->world; ++r)
            LR_NCCL(g_rccl.recv(static_cast<unsigned char*>(recv) + (uint64_t)r * bytes, bytes, ncclUint8, r, comm->comm, (hipStream_t)stream));
    LR_NCCL(g_rccl.group_end());
    return LR_OK;
}
int lr_allreduce_sum_f64(lr_comm* comm, double* buf, uint64_t count, void* stream) {
    if (!comm || !buf) return fail(LR_ERR_INVALID, "comm / buf is NULL");
    LR_HIP(hipSetDevice(comm->device));
    LR_NCCL(g_rccl.all_reduce(buf, buf, count, ncclDouble, ncclSum, comm->comm, (hipStream_t)stream));
    return LR_OK;
}

}  // extern "C"
