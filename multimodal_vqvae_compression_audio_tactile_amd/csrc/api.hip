// api.hip -- the C ABI of libmvq_hip.so (declared in include/mvq.h): argument checks and kernel dispatch.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mvq.h"
#include "conv_dispatch.hpp"
#include "kernels_small.hpp"

#include <map>
#include <string>
#include <vector>

namespace mvq {
// ---- per-launch HIP-event profiler ------------------------------------------------------------------------------------
namespace {
struct ProfRec { std::string name; double flops; hipEvent_t e0, e1; };
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;          // events are reused between sessions
bool g_prof_on = false;
hipError_t g_prof_err = hipSuccess;           // first hipEventCreate / hipEventRecord failure of the session (reported by _end)
hipEvent_t prof_event()
{
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) { if (g_prof_err == hipSuccess) g_prof_err = rc; return nullptr; }
    return e;
}
}  // namespace
bool prof_enabled() { return g_prof_on; }
// environment overrides a launcher has honoured (MVQ_NO_DMA, MVQ_ROWFAST_MAX_KB, MVQ_NO_TOKEN_RVQ): part of mvq_build_flags()
static unsigned g_env_flags = 0;
void note_env_override(unsigned bit) { __atomic_fetch_or(&g_env_flags, bit, __ATOMIC_RELAXED); }
int prof_begin(const char* kernel_name, double flops, hipStream_t s)
{
    if (!g_prof_on) return -1;
    ProfRec r{kernel_name, flops, prof_event(), nullptr};
    if (!r.e0) return -1;
    r.e1 = prof_event();
    if (!r.e1) { g_prof_pool.push_back(r.e0); return -1; }      // do not leak the first event when the second cannot be made
    const hipError_t rc = hipEventRecord(r.e0, s);
    if (rc != hipSuccess) {
        if (g_prof_err == hipSuccess) g_prof_err = rc;
        g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1);
        return -1;
    }
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}
void prof_end(int idx, hipStream_t s)
{
    if (idx < 0 || idx >= (int)g_prof.size()) return;
    const hipError_t rc = hipEventRecord(g_prof[idx].e1, s);
    if (rc != hipSuccess && g_prof_err == hipSuccess) g_prof_err = rc;
}
}  // namespace mvq

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
int hipfail(hipError_t e, const char* what)
{
    return fail(MVQ_EHIP, "%s: %s", what, hipGetErrorString(e));
}
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace
namespace mvq {
// stacks.hip reports through the same mvq_last_error() text
void set_last_error(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
}
namespace {
inline int conv_out_len(int tin, int ks, int stride, int dil, int pad)
{
    const int span = tin + 2 * pad - dil * (ks - 1) - 1;
    return span < 0 ? 0 : span / stride + 1;
}
}  // namespace

extern "C" {

int mvq_profile_begin(void)
{
    for (auto& r : mvq::g_prof) { mvq::g_prof_pool.push_back(r.e0); mvq::g_prof_pool.push_back(r.e1); }
    mvq::g_prof.clear();
    mvq::g_prof_err = hipSuccess;
    mvq::g_prof_on = true;
    return MVQ_OK;
}

int mvq_profile_reserve(int n_launches)
{
    /* create the event pairs of `n_launches` launches ahead of time, so that no hipEventCreate falls into a timed region */
    if (n_launches < 0) return fail(MVQ_EINVAL, "profile_reserve: negative count");
    std::vector<hipEvent_t> made;
    const size_t want = 2 * (size_t)n_launches;
    while (mvq::g_prof_pool.size() + made.size() < want) {
        hipEvent_t e = nullptr;
        const hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) { for (auto m : made) mvq::g_prof_pool.push_back(m); return hipfail(rc, "profile_reserve: hipEventCreate"); }
        made.push_back(e);
    }
    for (auto m : made) mvq::g_prof_pool.push_back(m);
    return MVQ_OK;
}

int mvq_profile_end(mvq_profile_entry* out, int max_entries, int* n_entries)
{
    return mvq_profile_end2(out, max_entries, n_entries, nullptr);
}

int mvq_profile_end2(mvq_profile_entry* out, int max_entries, int* n_entries, int* n_total)
{
    /* *n_entries = rows WRITTEN (<= max_entries); kernels beyond max_entries are dropped, never written past the buffer;
     * *n_total = instantiations the session saw (> *n_entries means the buffer truncated the table) */
    mvq::g_prof_on = false;
    if (n_total) *n_total = 0;
    if (!n_entries || max_entries < 0 || (max_entries > 0 && !out)) return fail(MVQ_EINVAL, "profile_end: bad argument");
    *n_entries = 0;
    auto recycle = [] {
        for (auto& r : mvq::g_prof) { mvq::g_prof_pool.push_back(r.e0); mvq::g_prof_pool.push_back(r.e1); }
        mvq::g_prof.clear();
    };
    if (mvq::g_prof_err != hipSuccess) {
        const hipError_t e = mvq::g_prof_err;
        mvq::g_prof_err = hipSuccess;
        recycle();
        return hipfail(e, "profile: an event could not be created or recorded during the session");
    }
    std::map<std::string, mvq_profile_entry> agg;
    for (auto& r : mvq::g_prof) {
        hipError_t e = hipEventSynchronize(r.e1);
        if (e != hipSuccess) { recycle(); return hipfail(e, "profile_end: hipEventSynchronize"); }
        float ms = 0.0f;
        e = hipEventElapsedTime(&ms, r.e0, r.e1);
        if (e != hipSuccess) { recycle(); return hipfail(e, "profile_end: hipEventElapsedTime"); }
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            mvq_profile_entry z{};
            snprintf(z.kernel, sizeof(z.kernel), "%s", r.name.c_str());
            it = agg.emplace(r.name, z).first;
        }
        it->second.seconds += 1e-3 * (double)ms;
        it->second.flops += r.flops;
        it->second.launches += 1;
    }
    recycle();
    int i = 0;
    for (auto& kv : agg) {
        if (i >= max_entries) break;
        out[i++] = kv.second;
    }
    *n_entries = i;
    if (n_total) *n_total = (int)agg.size();
    return MVQ_OK;
}

int mvq_abi_version(void) { return 3; }
unsigned mvq_build_flags(void)
{
    /* every conv translation unit is compiled with the same flags (one Makefile rule); also peek at the environment knobs so
     * that a process which has not launched anything yet already reports them */
    if (getenv("MVQ_NO_DMA")) mvq::note_env_override(MVQ_BF_ENV_NO_DMA);
    if (getenv("MVQ_ROWFAST_MAX_KB")) mvq::note_env_override(MVQ_BF_ENV_ROWFAST);
    if (getenv("MVQ_NO_TOKEN_RVQ")) mvq::note_env_override(MVQ_BF_ENV_NO_TOKEN_RVQ);
    if (getenv("MVQ_LN_TILE32")) mvq::note_env_override(0x800);
    if (getenv("MVQ_LAT_MAX_TILES")) mvq::note_env_override(MVQ_BF_ENV_LAT_TILES);
    if (getenv("MVQ_NO_DAC_RVQ_LAT")) mvq::note_env_override(0x2000);
    if (getenv("MVQ_NO_LN_LAT")) mvq::note_env_override(0x4000);
    if (getenv("MVQ_SMALL_TILE_MAX")) mvq::note_env_override(0x8000);
    if (getenv("MVQ_F16_NO192") || getenv("MVQ_F16_NO_WIDE")) mvq::note_env_override(0x20);      // tile A/B knobs of the opt-in f16x3 mode
    return mvq::conv_compile_flags() | __atomic_load_n(&mvq::g_env_flags, __ATOMIC_RELAXED);
}
const char* mvq_last_error(void) { return g_err; }

int mvq_device_query(int* cu_count, int* lds_bytes_per_cu, char* arch, int arch_len)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hipfail(e, "hipGetDevice");
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return hipfail(e, "hipGetDeviceProperties");
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return MVQ_OK;
}

int mvq_weight_norm_f32(const float* v, const float* g, float* w, int rows, int inner, void* stream)
{
    if (!v || !g || !w || rows <= 0 || inner <= 0) return fail(MVQ_EINVAL, "weight_norm: bad argument");
    hipError_t e = mvq::launch_weight_norm(v, g, w, rows, inner, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "weight_norm");
}

size_t mvq_conv1d_packed_floats(int cin, int cout, int ks)
{
    if (cin <= 0 || cout <= 0 || ks <= 0) return 0;
    return (size_t)cin * ks * mvq::conv_mpad(cout);
}

int mvq_conv1d_pack_f32(const float* w, float* wp, int cin, int cout, int ks, void* stream)
{
    if (!w || !wp || cin <= 0 || cout <= 0 || ks <= 0) return fail(MVQ_EINVAL, "conv1d_pack: bad argument");
    hipError_t e = mvq::launch_pack_conv1d(w, wp, cin, cout, ks, mvq::conv_mpad(cout), S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_pack");
}

size_t mvq_conv_transpose1d_packed_floats(int cin, int cout, int stride)
{
    if (cin <= 0 || cout <= 0 || stride <= 0) return 0;
    return (size_t)cin * 2 * mvq::conv_mpad(cout * stride);
}

int mvq_conv_transpose1d_pack_f32(const float* w, float* wp, int cin, int cout, int stride, void* stream)
{
    if (!w || !wp || cin <= 0 || cout <= 0 || stride <= 0) return fail(MVQ_EINVAL, "conv_transpose1d_pack: bad argument");
    hipError_t e = mvq::launch_pack_convtr(w, wp, cin, cout, stride, mvq::conv_mpad(cout * stride), S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv_transpose1d_pack");
}

static hipError_t dispatch_conv1d(const mvq::ConvArgs& a, int ks, int stride, int dil, hipStream_t s)
{
    const int bm = mvq::conv_tile_bm(a.Cout);
    hipError_t e = hipErrorInvalidValue;
    /* latency regime (one segment, a batch of six: conv_lat.hip): the 128-row tiling would put a block on fewer than 160 of the
     * 256 CUs and the launch has few enough 16 x 16 tiles -> one wave per tile on v_mfma_f32_16x16x4_f32, same fma chains */
    if (a.Cout >= 16 && a.Cin >= 32 && mvq::conv_underfilled(a) && mvq::conv_lat_wanted(a, ks)) {
        e = mvq::launch_conv_lat(a, ks, stride, dil, s);
        if (e != hipErrorInvalidValue) return e;
    }
    if (a.Cout >= 32 && a.Cin >= 32) {                 // a dense (channels x kernel) tile exists
        if (ks == 7 && stride == 1 && a.Cin % 8 == 0) e = mvq::launch_conv_k7(a, dil, bm, s);
        else if (ks == 1 && stride == 1 && dil == 1 && a.Cin % 32 == 0) e = mvq::launch_conv_k1k3(a, 1, bm, s);
        else if (ks == 3 && stride == 1 && dil == 1 && a.Cin % 16 == 0) e = mvq::launch_conv_k1k3(a, 3, bm, s);
        else if (ks == 2 * stride && dil == 1 && a.Cin % 16 == 0) e = mvq::launch_conv_strided(a, stride, bm, s);
    }
    return e;
}

static hipError_t dispatch_convtr(const mvq::ConvArgs& a, hipStream_t s)
{
    if (a.Cin % 32 != 0 || a.Mrows < 64) return hipErrorInvalidValue;
    return mvq::launch_conv_tr(a, mvq::conv_tile_bm(a.Mrows), s);
}

/* ---- opt-in bf16x6 arithmetic mode (conv_k7_bf16.hip; include/mvq.h) ---- */
size_t mvq_bf16x3_split_bytes(int batch, int c, int t)
{
    if (batch <= 0 || c <= 0 || t <= 0) return 0;
    return (size_t)batch * c * t * 6;
}
int mvq_bf16x3_split_f32(const float* x, void* xs, int batch, int c, int t, void* stream)
{
    if (batch < 0 || c <= 0 || t < 0 || c % 8 != 0) return fail(MVQ_EINVAL, "bf16x3_split: bad shape B=%d C=%d T=%d (C %% 8 == 0)", batch, c, t);
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!x || !xs) return fail(MVQ_EINVAL, "bf16x3_split: null tensor");
    if ((reinterpret_cast<uintptr_t>(xs) & 15) != 0) return fail(MVQ_EINVAL, "bf16x3_split: xs must be 16-byte aligned");
    const hipError_t e = mvq::launch_bf16x3_split(x, xs, batch, c, t, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "bf16x3_split");
}
size_t mvq_conv1d_k7_bf16x3_packed_bytes(int cout, int cin)
{
    if (cout <= 0 || cin <= 0 || mvq::bf16x6_tile_rows(cout) == 0 || cin % 16 != 0) return 0;
    return (size_t)cout * cin * 7 * 6;
}
int mvq_conv1d_k7_pack_bf16x3(const float* w, void* wq, int cout, int cin, int dgrad, void* stream)
{
    if (cout <= 0 || cin <= 0 || mvq::bf16x6_tile_rows(cout) == 0 || cin % 16 != 0)
        return fail(MVQ_EINVAL, "conv1d_k7_pack_bf16x3: Cout %d must be a multiple of 128 or 96 and Cin %d of 16", cout, cin);
    if (!w || !wq) return fail(MVQ_EINVAL, "conv1d_k7_pack_bf16x3: null tensor");
    if ((reinterpret_cast<uintptr_t>(wq) & 15) != 0) return fail(MVQ_EINVAL, "conv1d_k7_pack_bf16x3: wq must be 16-byte aligned");
    const hipError_t e = mvq::launch_bf16x3_pack_k7(w, wq, cout, cin, dgrad != 0, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_k7_pack_bf16x3");
}
int mvq_conv1d_k7_bf16x6_f32(const void* xs, const void* wq, const float* bias, const float* alpha_out, float* y, float* y2,
                             const float* dsn_src, const float* dsn_alpha, const float* residual,
                             int batch, int cin, int t, int cout, int dil, int tvalid, void* stream)
{
    if (batch < 0 || cin <= 0 || cout <= 0 || t < 0 || mvq::bf16x6_tile_rows(cout) == 0 || cin % 16 != 0)
        return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: bad shape B=%d Cin=%d T=%d Cout=%d (Cout %% 128 or 96, Cin %% 16)", batch, cin, t, cout);
    if (dil != 1 && dil != 3 && dil != 9) return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: dilation %d not in {1, 3, 9}", dil);
    if (tvalid < 0 || tvalid > t) return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: tvalid %d outside [0, %d]", tvalid, t);
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!xs || !wq || !y) return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: null tensor");
    if (((reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(wq)) & 15) != 0)
        return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: xs / wq must be 16-byte aligned");
    if ((long long)batch * ((t + 127) / 128) > 0x7fffffffLL) return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: grid too large");
    if (y2 && !alpha_out) return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: the dual output needs alpha_out");
    if ((dsn_src != nullptr) != (dsn_alpha != nullptr)) return fail(MVQ_EINVAL, "conv1d_k7_bf16x6: dsn_src and dsn_alpha go together");
    mvq::K7Extra ex; ex.y2 = y2; ex.dsn_src = dsn_src; ex.dsn_alpha = dsn_alpha; ex.residual = residual;
    const hipError_t e = mvq::launch_conv_k7_bf16x6(xs, wq, bias, alpha_out, y, batch, cin, t, cout, dil, tvalid, ex, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_k7_bf16x6");
}

/* f16x3 form of the opt-in mode (include/mvq.h) */
int mvq_f16x2_split_f32(const float* x, void* xs, uint32_t* xamax, int batch, int c, int t, void* stream)
{
    if (batch < 0 || c <= 0 || t < 0 || c % 8 != 0) return fail(MVQ_EINVAL, "f16x2_split: bad shape B=%d C=%d T=%d (C %% 8 == 0)", batch, c, t);
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!x || !xs || !xamax) return fail(MVQ_EINVAL, "f16x2_split: null tensor");
    if ((reinterpret_cast<uintptr_t>(xs) & 15) != 0) return fail(MVQ_EINVAL, "f16x2_split: xs must be 16-byte aligned");
    const hipError_t e = mvq::launch_f16x2_split(x, xs, xamax, batch, c, t, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "f16x2_split");
}
size_t mvq_conv1d_k7_f16x2_packed_bytes(int cout, int cin)
{
    if (cout <= 0 || cin <= 0 || mvq::bf16x6_tile_rows(cout) == 0 || cin % 16 != 0) return 0;
    return (size_t)cout * cin * 7 * 4;
}
int mvq_conv1d_k7_pack_f16x2(const float* w, void* wq, uint32_t* wamax, int cout, int cin, int dgrad, void* stream)
{
    if (cout <= 0 || cin <= 0 || mvq::bf16x6_tile_rows(cout) == 0 || cin % 16 != 0)
        return fail(MVQ_EINVAL, "conv1d_k7_pack_f16x2: Cout %d must be a multiple of 128 or 96 and Cin %d of 16", cout, cin);
    if (!w || !wq || !wamax) return fail(MVQ_EINVAL, "conv1d_k7_pack_f16x2: null tensor");
    if ((reinterpret_cast<uintptr_t>(wq) & 15) != 0) return fail(MVQ_EINVAL, "conv1d_k7_pack_f16x2: wq must be 16-byte aligned");
    const hipError_t e = mvq::launch_f16x2_pack_k7(w, wq, wamax, cout, cin, dgrad != 0, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_k7_pack_f16x2");
}
int mvq_conv1d_k7_f16x3_f32(const void* xs, const uint32_t* xamax, const void* wq, const uint32_t* wamax, const float* bias,
                            const float* alpha_out, float* y, float* y2, const float* dsn_src, const float* dsn_alpha,
                            const float* residual, int batch, int cin, int t, int cout, int dil, int tvalid, void* stream)
{
    if (batch < 0 || cin <= 0 || cout <= 0 || t < 0 || mvq::bf16x6_tile_rows(cout) == 0 || cin % 16 != 0)
        return fail(MVQ_EINVAL, "conv1d_k7_f16x3: bad shape B=%d Cin=%d T=%d Cout=%d (Cout %% 128 or 96, Cin %% 16)", batch, cin, t, cout);
    if (dil != 1 && dil != 3 && dil != 9) return fail(MVQ_EINVAL, "conv1d_k7_f16x3: dilation %d not in {1, 3, 9}", dil);
    if (tvalid < 0 || tvalid > t) return fail(MVQ_EINVAL, "conv1d_k7_f16x3: tvalid %d outside [0, %d]", tvalid, t);
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!xs || !wq || !y || !xamax || !wamax) return fail(MVQ_EINVAL, "conv1d_k7_f16x3: null tensor");
    if (((reinterpret_cast<uintptr_t>(xs) | reinterpret_cast<uintptr_t>(wq)) & 15) != 0)
        return fail(MVQ_EINVAL, "conv1d_k7_f16x3: xs / wq must be 16-byte aligned");
    if ((long long)batch * ((t + 127) / 128) > 0x7fffffffLL) return fail(MVQ_EINVAL, "conv1d_k7_f16x3: grid too large");
    if (y2 && !alpha_out) return fail(MVQ_EINVAL, "conv1d_k7_f16x3: the dual output needs alpha_out");
    if ((dsn_src != nullptr) != (dsn_alpha != nullptr)) return fail(MVQ_EINVAL, "conv1d_k7_f16x3: dsn_src and dsn_alpha go together");
    mvq::K7Extra ex; ex.y2 = y2; ex.dsn_src = dsn_src; ex.dsn_alpha = dsn_alpha; ex.residual = residual;
    const hipError_t e = mvq::launch_conv_k7_f16x3(xs, xamax, wq, wamax, bias, alpha_out, y, batch, cin, t, cout, dil, tvalid, ex, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_k7_f16x3");
}

/* which kernel instantiation mvq_conv1d_f32 / mvq_conv_transpose1d_f32 launches for a shape (profiling aid):
 * runs the real dispatch in "name mode" */
int mvq_conv_kernel_name(int batch, int cin, int cout, int ks, int stride, int dil, int transposed, int tin, char* buf, int len)
{
    if (!buf || len <= 0) return fail(MVQ_EINVAL, "conv_kernel_name: bad buffer");
    mvq::ConvArgs a{};
    a.B = batch; a.Cin = cin; a.Tin = tin; a.Cout = cout; a.name_out = buf; a.name_len = len;
    hipError_t e;
    if (transposed) {
        a.Mrows = cout * stride; a.Mpad = mvq::conv_mpad(a.Mrows); a.Ncols = tin + 1; a.up_s = stride; a.pad = 1;
        e = dispatch_convtr(a, nullptr);
        if (e != hipSuccess) snprintf(buf, len, "convtr_direct_kernel");
    } else {
        const int pad = stride == 1 ? (ks - 1) * dil / 2 : (stride + 1) / 2;
        a.Mrows = cout; a.Mpad = mvq::conv_mpad(cout); a.Ncols = conv_out_len(tin, ks, stride, dil, pad); a.pad = pad;
        a.Tout = a.Ncols; a.up_s = 1;
        e = dispatch_conv1d(a, ks, stride, dil, nullptr);
        if (e != hipSuccess)
            snprintf(buf, len, "%s", (cin == 1 && ks == 7 && stride == 1 && dil == 1) ? "conv1d_cin1_kernel<7>"
                                   : (cout == 1 && ks == 7 && stride == 1 && dil == 1) ? "conv1d_cout1_kernel<7>"
                                                                                       : "conv1d_direct_kernel");
    }
    return MVQ_OK;
}

int mvq_conv1d_dual_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                        const float* residual, const float* alpha_out, float* y, float* y2, const float* alpha2,
                        int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act, void* stream);

int mvq_conv1d_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                   const float* residual, const float* alpha_out, float* y,
                   int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act, void* stream)
{
    return mvq_conv1d_dual_f32(x, wp, bias, alpha_in, residual, alpha_out, y, nullptr, nullptr,
                               batch, cin, tin, cout, ks, stride, dil, pad, act, stream);
}

int mvq_conv1d_dual_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                        const float* residual, const float* alpha_out, float* y, float* y2, const float* alpha2,
                        int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act, void* stream)
{
    return mvq_conv1d_padded_f32(x, wp, bias, alpha_in, residual, alpha_out, y, y2, alpha2, batch, cin, tin, cout, ks, stride,
                                 dil, pad, act, 0, stream);
}

int mvq_conv1d_padded_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                          const float* residual, const float* alpha_out, float* y, float* y2, const float* alpha2,
                          int batch, int cin, int tin, int cout, int ks, int stride, int dil, int pad, int act, int tvalid,
                          void* stream)
{
    if (batch < 0 || cin <= 0 || cout <= 0 || tin < 0 || ks <= 0 || stride <= 0 || dil <= 0 || pad < 0)
        return fail(MVQ_EINVAL, "conv1d: bad shape B=%d Cin=%d Tin=%d Cout=%d ks=%d s=%d d=%d p=%d", batch, cin, tin, cout, ks, stride, dil, pad);
    if (act != MVQ_ACT_NONE && act != MVQ_ACT_TANH && act != MVQ_ACT_GELU) return fail(MVQ_EINVAL, "conv1d: bad act %d", act);
    const int tout = conv_out_len(tin, ks, stride, dil, pad);
    if (batch == 0 || tout == 0) return MVQ_OK;
    if (!x || !wp || !y) return fail(MVQ_EINVAL, "conv1d: null tensor");
    if ((y2 != nullptr) != (alpha2 != nullptr)) return fail(MVQ_EINVAL, "conv1d: y2 and alpha2 go together");
    if (tvalid < 0 || tvalid > tout) return fail(MVQ_EINVAL, "conv1d: tvalid %d outside [0, %d]", tvalid, tout);
    const int mpad = mvq::conv_mpad(cout);

    mvq::ConvArgs a{};
    a.x = x; a.wp = wp; a.bias = bias; a.alpha_in = alpha_in; a.residual = residual; a.alpha_out = alpha_out; a.y = y;
    a.B = batch; a.Cin = cin; a.Tin = tin; a.Cout = cout; a.Tout = tout; a.pad = pad; a.Mpad = mpad;
    a.Mrows = cout; a.Ncols = tout; a.act = act; a.up_s = 1; a.up_p = 0; a.y2 = y2; a.alpha2 = alpha2;
    a.tvalid = (tvalid == tout) ? 0 : tvalid;

    hipError_t e = dispatch_conv1d(a, ks, stride, dil, S(stream));
    if (e == hipErrorInvalidValue && (a.tvalid || act == MVQ_ACT_GELU))
        return fail(MVQ_EUNSUPPORTED, "conv1d: zero-padded rows / the GELU epilogue need an MFMA-tiled shape");
    if (e == hipErrorInvalidValue) {
        (void)hipGetLastError();
        mvq::DirectConvArgs d{x, wp, bias, alpha_in, residual, alpha_out, y, batch, cin, tin, cout, tout, ks, stride, dil, pad, mpad, act, y2, alpha2, nullptr, nullptr};
        e = mvq::launch_conv1d_direct(d, S(stream));
    }
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d");
}

static unsigned magic_div(int d) { return (unsigned)((((unsigned long long)1 << 32) + (unsigned)d - 1) / (unsigned)d); }   /* umulhi(n, magic) == n / d for n * d < 2^32 */

int mvq_conv1d_packed_rows_f32(const float* x, const float* wp, const float* bias, const float* residual, const float* alpha_out,
                               float* y, float* y2, const float* alpha2, int rows, int cin, int cout, int ks, int dil, int pad,
                               int act, int seg_per_row, int seg_period, int seg_valid, void* stream)
{
    if (rows < 0 || cin <= 0 || cout <= 0 || ks <= 0 || dil <= 0 || pad < 0 || seg_per_row <= 0 || seg_period <= 0 || seg_valid < 0)
        return fail(MVQ_EINVAL, "conv1d_packed_rows: bad shape");
    if (seg_period % 4 != 0 || seg_valid > seg_period || (ks - 1) * dil != 2 * pad || pad > seg_period - seg_valid)
        return fail(MVQ_EINVAL, "conv1d_packed_rows: needs a 'same' stride-1 conv, a period that is a multiple of 4 and a gap of at "
                                "least `pad` zero columns between segments (period %d, valid %d, pad %d)", seg_period, seg_valid, pad);
    if (act != MVQ_ACT_NONE && act != MVQ_ACT_TANH && act != MVQ_ACT_GELU) return fail(MVQ_EINVAL, "conv1d_packed_rows: bad act %d", act);
    const long long t = (long long)seg_per_row * seg_period;
    if (t * seg_period >= ((long long)1 << 32)) return fail(MVQ_EINVAL, "conv1d_packed_rows: row too long");
    if (rows == 0 || seg_valid == 0) return MVQ_OK;
    if (!x || !wp || !y) return fail(MVQ_EINVAL, "conv1d_packed_rows: null tensor");
    if ((y2 != nullptr) != (alpha2 != nullptr)) return fail(MVQ_EINVAL, "conv1d_packed_rows: y2 and alpha2 go together");
    mvq::ConvArgs a{};
    a.x = x; a.wp = wp; a.bias = bias; a.residual = residual; a.alpha_out = alpha_out; a.y = y; a.y2 = y2; a.alpha2 = alpha2;
    a.B = rows; a.Cin = cin; a.Tin = (int)t; a.Cout = cout; a.Tout = (int)t; a.pad = pad; a.Mpad = mvq::conv_mpad(cout);
    a.Mrows = cout; a.Ncols = (int)t; a.act = act; a.up_s = 1;
    a.tper = seg_period; a.tper_valid = seg_valid; a.tper_magic = magic_div(seg_period);
    hipError_t e = dispatch_conv1d(a, ks, 1, dil, S(stream));
    if (e == hipErrorInvalidValue) return fail(MVQ_EUNSUPPORTED, "conv1d_packed_rows: needs an MFMA-tiled shape");
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_packed_rows");
}

int mvq_conv1d_vpacked_f32(const float* x, const float* wp, const float* bias, const float* residual, const float* alpha_out,
                           float* y, float* y2, const float* alpha2, int batch, int cin, int tin_rows, int tin_valid, int cout,
                           int ks, int stride, int dil, int pad, int act, int seg_per_row, int per_in, int tout_rows, void* stream)
{
    if (batch < 0 || cin <= 0 || cout <= 0 || ks <= 0 || stride <= 0 || dil <= 0 || pad < 0 || tin_rows <= 0 || tin_valid < 0 ||
        tin_valid > tin_rows || seg_per_row <= 0 || per_in <= 0 || tout_rows <= 0)
        return fail(MVQ_EINVAL, "conv1d_vpacked: bad shape");
    if (act != MVQ_ACT_NONE && act != MVQ_ACT_TANH && act != MVQ_ACT_GELU) return fail(MVQ_EINVAL, "conv1d_vpacked: bad act %d", act);
    const int tout = conv_out_len(tin_valid, ks, stride, dil, pad);             /* valid outputs per item */
    /* geometry of the virtual row: the output period is the physical output row, the input period is stride x that; between the
     * data of neighbouring items there must be at least `pad` zeros on the left and the last valid output's overhang on the right
     * (columns [tin_valid, tin_rows) of x are the caller's zero tail, columns beyond tin_rows come from the zero block) */
    const int per_out = tout_rows;
    const int overhang = (tout > 0 ? (tout - 1) * stride - pad + dil * (ks - 1) : 0) - (tin_valid - 1);   /* zeros needed behind the data */
    if (tin_rows % 4 != 0 || tout_rows % 4 != 0 || per_in % 4 != 0 || per_in != stride * per_out || tout > tout_rows ||
        tin_rows > per_in || per_in - tin_valid < pad || per_in - tin_valid < overhang)
        return fail(MVQ_EINVAL, "conv1d_vpacked: needs 16-byte rows, per_in == stride * tout_rows, tout <= tout_rows and a gap of at least "
                                "max(pad, overhang) zeros between items (tin_rows %d valid %d per_in %d tout %d tout_rows %d pad %d overhang %d)",
                    tin_rows, tin_valid, per_in, tout, tout_rows, pad, overhang);
    if ((long long)seg_per_row * per_in * per_in >= ((long long)1 << 32)) return fail(MVQ_EINVAL, "conv1d_vpacked: virtual row too long");
    if (batch == 0 || tout == 0) return MVQ_OK;
    if (!x || !wp || !y) return fail(MVQ_EINVAL, "conv1d_vpacked: null tensor");
    if ((y2 != nullptr) != (alpha2 != nullptr)) return fail(MVQ_EINVAL, "conv1d_vpacked: y2 and alpha2 go together");
    if (y2 && alpha_out) return fail(MVQ_EUNSUPPORTED, "conv1d_vpacked: a dual output and an output Snake together are not supported");
    mvq::ConvArgs a{};
    a.x = x; a.wp = wp; a.bias = bias; a.residual = residual; a.alpha_out = alpha_out; a.y = y; a.y2 = y2; a.alpha2 = alpha2;
    a.B = (batch + seg_per_row - 1) / seg_per_row;                               /* virtual rows */
    a.Cin = cin; a.Tin = seg_per_row * per_in; a.Cout = cout; a.Tout = tout_rows; a.pad = pad; a.Mpad = mvq::conv_mpad(cout);
    a.Mrows = cout; a.Ncols = seg_per_row * per_out; a.act = act; a.up_s = 1;
    a.vp_seg = seg_per_row; a.vp_per_in = per_in; a.vp_valid_in = tin_rows; a.vp_tin_phys = tin_rows; a.vp_btrue = batch;
    a.vp_per_out = per_out; a.vp_valid_out = tout; a.vp_magic_in = magic_div(per_in); a.vp_magic_out = magic_div(per_out);
    hipError_t e = dispatch_conv1d(a, ks, stride, dil, S(stream));
    if (e == hipErrorInvalidValue) return fail(MVQ_EUNSUPPORTED, "conv1d_vpacked: needs an MFMA-tiled, LDS-DMA-staged shape with the regular epilogue");
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_vpacked");
}

int mvq_conv_transpose1d_packed_rows_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                         const float* alpha_out, float* y, float* y2, const float* alpha2, int rows, int cin,
                                         int cout, int stride, int pad, int seg_per_row, int seg_period, int seg_valid,
                                         int batch_out, void* stream)
{
    if (rows < 0 || cin <= 0 || cout <= 0 || stride <= 0 || pad < 0 || seg_per_row <= 0 || seg_period <= 0 || seg_valid <= 0 || batch_out < 0)
        return fail(MVQ_EINVAL, "conv_transpose1d_packed_rows: bad shape");
    /* a segment's outputs must not reach into its neighbour's: the first `pad` positions are dropped, and the last outputs come
     * from input column seg_valid (a zero of the gap) -- which has to exist */
    if (seg_valid >= seg_period || batch_out > rows * seg_per_row)
        return fail(MVQ_EINVAL, "conv_transpose1d_packed_rows: needs at least one zero column between segments and rows * seg_per_row >= batch_out");
    const long long tin = (long long)seg_per_row * seg_period;
    const int tout = (seg_valid - 1) * stride - 2 * pad + 2 * stride;             /* per segment, as the unpacked layer gives */
    const long long per_out = (long long)seg_period * stride;
    if (tout <= 0 || tout > per_out) return fail(MVQ_EINVAL, "conv_transpose1d_packed_rows: bad segment length");
    if (tin * stride * per_out >= ((long long)1 << 32)) return fail(MVQ_EINVAL, "conv_transpose1d_packed_rows: row too long");
    if (rows == 0 || batch_out == 0) return MVQ_OK;
    if (!x || !wp || !y) return fail(MVQ_EINVAL, "conv_transpose1d_packed_rows: null tensor");
    if ((y2 != nullptr) != (alpha2 != nullptr)) return fail(MVQ_EINVAL, "conv_transpose1d_packed_rows: y2 and alpha2 go together");
    const int mrows = cout * stride;
    mvq::ConvArgs a{};
    a.x = x; a.wp = wp; a.bias = bias; a.alpha_in = alpha_in; a.alpha_out = alpha_out; a.y = y; a.y2 = y2; a.alpha2 = alpha2;
    a.B = rows; a.Cin = cin; a.Tin = (int)tin; a.Cout = cout; a.Tout = tout; a.pad = 1; a.Mpad = mvq::conv_mpad(mrows);
    a.Mrows = mrows; a.Ncols = (int)tin;          /* no boundary column: column tin would only feed positions past the last segment */
    a.up_s = stride; a.up_p = pad;
    a.up_per_out = (int)per_out; a.up_valid_out = tout; a.up_seg = seg_per_row; a.up_btrue = batch_out; a.up_magic = magic_div((int)per_out);
    hipError_t e = dispatch_convtr(a, S(stream));
    if (e == hipErrorInvalidValue) return fail(MVQ_EUNSUPPORTED, "conv_transpose1d_packed_rows: needs an MFMA-tiled shape");
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv_transpose1d_packed_rows");
}

static bool ru_fusable(int c, int dil)
{
    return (c == 64 || c == 96 || c == 128) && (dil == 1 || dil == 3 || dil == 9);
}

int mvq_residual_unit_kernel_name(int c, int dil, char* buf, int len)
{
    if (!buf || len <= 0) return fail(MVQ_EINVAL, "residual_unit_kernel_name: bad buffer");
    if (!ru_fusable(c, dil)) { snprintf(buf, len, "%s", "(two launches)"); return MVQ_OK; }
    mvq::ConvArgs a{};
    a.Cin = c; a.Cout = c; a.Mpad = mvq::conv_mpad(c); a.name_out = buf; a.name_len = len;
    (void)mvq::launch_residual_unit_fused(a, dil, nullptr);
    return MVQ_OK;
}

size_t mvq_residual_unit_scratch_floats(int batch, int c, int t, int dil)
{
    if (batch <= 0 || c <= 0 || t <= 0) return 0;
    return ru_fusable(c, dil) ? 0 : (size_t)batch * c * t;
}

int mvq_residual_unit_dual_f32(const float* x, const float* x_snaked, const float* w7p, const float* b7,
                               const float* alpha_a, const float* alpha_b, const float* w1p, const float* b1,
                               const float* alpha_next, float* y, float* y2, const float* alpha2, float* scratch,
                               int batch, int c, int t, int dil, void* stream);

int mvq_residual_unit_f32(const float* x, const float* w7p, const float* b7, const float* alpha_a,
                          const float* alpha_b, const float* w1p, const float* b1, const float* alpha_next,
                          float* y, float* scratch, int batch, int c, int t, int dil, void* stream)
{
    return mvq_residual_unit_dual_f32(x, nullptr, w7p, b7, alpha_a, alpha_b, w1p, b1, alpha_next, y, nullptr, nullptr,
                                      scratch, batch, c, t, dil, stream);
}

int mvq_residual_unit_dual_f32(const float* x, const float* x_snaked, const float* w7p, const float* b7,
                               const float* alpha_a, const float* alpha_b, const float* w1p, const float* b1,
                               const float* alpha_next, float* y, float* y2, const float* alpha2, float* scratch,
                               int batch, int c, int t, int dil, void* stream)
{
    return mvq_residual_unit_padded_f32(x, x_snaked, w7p, b7, alpha_a, alpha_b, w1p, b1, alpha_next, y, y2, alpha2, scratch,
                                        batch, c, t, dil, 0, stream);
}

int mvq_residual_unit_padded_f32(const float* x, const float* x_snaked, const float* w7p, const float* b7,
                                 const float* alpha_a, const float* alpha_b, const float* w1p, const float* b1,
                                 const float* alpha_next, float* y, float* y2, const float* alpha2, float* scratch,
                                 int batch, int c, int t, int dil, int tvalid, void* stream)
{
    if (batch < 0 || c <= 0 || t < 0 || dil <= 0) return fail(MVQ_EINVAL, "residual_unit: bad shape");
    if (tvalid < 0 || tvalid > t) return fail(MVQ_EINVAL, "residual_unit: tvalid outside [0, t]");
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!x || !w7p || !w1p || !alpha_a || !alpha_b || !y) return fail(MVQ_EINVAL, "residual_unit: null tensor");
    if ((y2 != nullptr) != (alpha2 != nullptr)) return fail(MVQ_EINVAL, "residual_unit: y2 and alpha2 go together");
    if (ru_fusable(c, dil)) {
        mvq::ConvArgs a{};
        /* x_snaked = snake_a(x) from the producer's dual output: the unit stages it as is (LDS-DMA ring, no Snake on load);
         * the skip path still adds the RAW x in the epilogue */
        a.x = x_snaked ? x_snaked : x; a.wp = w7p; a.bias = b7; a.alpha_in = x_snaked ? nullptr : alpha_a; a.residual = x;
        a.alpha_out = alpha_next; a.y = y;
        a.B = batch; a.Cin = c; a.Tin = t; a.Cout = c; a.Tout = t; a.pad = 3 * dil; a.Mpad = mvq::conv_mpad(c);
        a.Mrows = c; a.Ncols = t; a.act = 0; a.up_s = 1; a.up_p = 0;
        a.alpha_mid = alpha_b; a.w2p = w1p; a.bias2 = b1; a.y2 = y2; a.alpha2 = alpha2;
        a.tvalid = (tvalid == t) ? 0 : tvalid;
        hipError_t e = mvq::launch_residual_unit_fused(a, dil, S(stream));
        return e == hipSuccess ? MVQ_OK : hipfail(e, "residual_unit(fused)");
    }
    if (!scratch) return fail(MVQ_EINVAL, "residual_unit: scratch required for C=%d (see mvq_residual_unit_scratch_floats)", c);
    int rc = mvq_conv1d_padded_f32(x_snaked ? x_snaked : x, w7p, b7, x_snaked ? nullptr : alpha_a, nullptr, alpha_b, scratch,
                                   nullptr, nullptr, batch, c, t, c, 7, 1, dil, 3 * dil, MVQ_ACT_NONE, tvalid, stream);
    if (rc != MVQ_OK) return rc;
    return mvq_conv1d_padded_f32(scratch, w1p, b1, nullptr, x, alpha_next, y, y2, alpha2, batch, c, t, c, 1, 1, 1, 0,
                                 MVQ_ACT_NONE, tvalid, stream);
}

int mvq_conv_transpose1d_dual_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                  const float* alpha_out, float* y, float* y2, const float* alpha2,
                                  int batch, int cin, int tin, int cout, int stride, int pad, void* stream);

int mvq_conv_transpose1d_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                             const float* alpha_out, float* y,
                             int batch, int cin, int tin, int cout, int stride, int pad, void* stream)
{
    return mvq_conv_transpose1d_dual_f32(x, wp, bias, alpha_in, alpha_out, y, nullptr, nullptr, batch, cin, tin, cout, stride, pad, stream);
}

int mvq_conv_transpose1d_dual_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                  const float* alpha_out, float* y, float* y2, const float* alpha2,
                                  int batch, int cin, int tin, int cout, int stride, int pad, void* stream)
{
    return mvq_conv_transpose1d_padded_f32(x, wp, bias, alpha_in, alpha_out, y, y2, alpha2, batch, cin, tin, cout, stride, pad,
                                           0, 0, stream);
}

int mvq_conv_transpose1d_padded_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                    const float* alpha_out, float* y, float* y2, const float* alpha2,
                                    int batch, int cin, int tin, int cout, int stride, int pad, int tout_rows, int tvalid,
                                    void* stream)
{
    return mvq_conv_transpose1d_op_f32(x, wp, bias, alpha_in, alpha_out, y, y2, alpha2, batch, cin, tin, cout, stride, pad, 0,
                                       tout_rows, tvalid, stream);
}

int mvq_conv_transpose1d_op_f32(const float* x, const float* wp, const float* bias, const float* alpha_in,
                                const float* alpha_out, float* y, float* y2, const float* alpha2,
                                int batch, int cin, int tin, int cout, int stride, int pad, int output_padding, int tout_rows,
                                int tvalid, void* stream)
{
    if (batch < 0 || cin <= 0 || cout <= 0 || tin < 0 || stride <= 0 || pad < 0)
        return fail(MVQ_EINVAL, "conv_transpose1d: bad shape");
    /* torch: output_padding < stride; the polyphase form reaches `pad` columns past the un-padded length, so <= pad as well */
    if (output_padding < 0 || output_padding >= stride || output_padding > pad)
        return fail(MVQ_EINVAL, "conv_transpose1d: output_padding %d outside [0, min(stride - 1, pad)]", output_padding);
    const int tnat = (tin - 1) * stride - 2 * pad + 2 * stride + output_padding;
    /* tout_rows: length of the output rows (0 = the natural length).  Shorter than natural: the input carries a zero tail and
     * only its true outputs are wanted.  Up to `pad` longer (the polyphase form visits those columns too): rows padded to a
     * multiple of 4, columns >= tvalid zeroed. */
    const int tout = tout_rows > 0 ? tout_rows : tnat;
    if (tout > tnat - output_padding + pad || tvalid < 0 || tvalid > tout || (tout > tnat && (tvalid == 0 || tvalid > tnat)))
        return fail(MVQ_EINVAL, "conv_transpose1d: tout_rows %d / tvalid %d inconsistent with the natural length %d", tout, tvalid, tnat);
    if (batch == 0 || tin == 0 || tout <= 0) return MVQ_OK;
    if (!x || !wp || !y) return fail(MVQ_EINVAL, "conv_transpose1d: null tensor");
    if ((y2 != nullptr) != (alpha2 != nullptr)) return fail(MVQ_EINVAL, "conv_transpose1d: y2 and alpha2 go together");
    const int mrows = cout * stride;
    const int mpad = mvq::conv_mpad(mrows);
    hipError_t e;
    {
        mvq::ConvArgs a{};
        a.x = x; a.wp = wp; a.bias = bias; a.alpha_in = alpha_in; a.residual = nullptr; a.alpha_out = alpha_out; a.y = y;
        a.B = batch; a.Cin = cin; a.Tin = tin; a.Cout = cout; a.Tout = tout; a.pad = 1; a.Mpad = mpad;
        a.Mrows = mrows; a.Ncols = tin + 1; a.act = 0; a.up_s = stride; a.up_p = pad; a.y2 = y2; a.alpha2 = alpha2;
        a.tvalid = (tvalid == tout) ? 0 : tvalid;
        e = dispatch_convtr(a, S(stream));
    }
    if (e == hipErrorInvalidValue && (tvalid || tout_rows)) return fail(MVQ_EUNSUPPORTED, "conv_transpose1d: zero-padded rows need an MFMA-tiled shape");
    if (e == hipErrorInvalidValue) {
        (void)hipGetLastError();
        mvq::DirectConvArgs d{x, wp, bias, alpha_in, nullptr, alpha_out, y, batch, cin, tin, cout, tout, 2 * stride, stride, 1, pad, mpad, 0, y2, alpha2, nullptr, nullptr};
        e = mvq::launch_convtr_direct(d, S(stream));
    }
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv_transpose1d");
}

int mvq_rvq_ema_forward_f32(const float* z, const float* books, float* q_out, int32_t* idx_out,
                            int batch, int dim, int t, int nb_use, int k, void* stream)
{
    if (batch < 0 || t < 0 || dim <= 0 || dim > 128 || dim % 4 != 0 || nb_use < 0 || k <= 0)
        return fail(MVQ_EINVAL, "rvq_ema_forward: bad shape B=%d D=%d T=%d nb=%d K=%d", batch, dim, t, nb_use, k);
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!z || !q_out || (nb_use > 0 && !books)) return fail(MVQ_EINVAL, "rvq_ema_forward: null tensor");
    hipError_t e = mvq::launch_rvq_ema_forward(z, books, q_out, idx_out, batch, dim, t, nb_use, k, 1, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "rvq_ema_forward");
}

size_t mvq_rvq_ema_step_scratch_bytes(int batch, int t, int nb, int k, int dim)
{
    if (batch <= 0 || t <= 0 || nb <= 0 || k <= 0 || dim <= 0) return 0;
    const size_t n = (size_t)batch * t;
    /* the per-book assignments, then the counting-sort workspace of the update (16-byte aligned) */
    return (((size_t)nb * n * sizeof(int32_t) + 15) & ~(size_t)15) + mvq::ema_update_scratch_bytes((int)n, nb, k, dim);
}

int mvq_rvq_ema_step_f32(const float* z_tokens, float* books, void* scratch,
                         int batch, int dim, int t, int nb, int k, float decay, void* stream)
{
    if (batch < 0 || t < 0 || dim <= 0 || dim > 128 || dim % 4 != 0 || nb <= 0 || k <= 0) return fail(MVQ_EINVAL, "rvq_ema_step: bad shape");
    if ((long long)batch * t >= (1ll << 24)) return fail(MVQ_EUNSUPPORTED, "rvq_ema_step: at most 2^24 - 1 tokens per call (counts are kept exact in fp32)");
    if (k > 16384) return fail(MVQ_EUNSUPPORTED, "rvq_ema_step: K <= 16384 (the per-segment histogram lives in LDS)");
    if (batch * t == 0) return MVQ_OK;
    if (!z_tokens || !books || !scratch) return fail(MVQ_EINVAL, "rvq_ema_step: null tensor");
    int32_t* idx = reinterpret_cast<int32_t*>(scratch);
    const size_t n = (size_t)batch * t;
    void* work = reinterpret_cast<char*>(scratch) + (((size_t)nb * n * sizeof(int32_t) + 15) & ~(size_t)15);
    hipError_t e = mvq::launch_rvq_ema_forward(z_tokens, books, nullptr, idx, batch, dim, t, nb, k, 0, S(stream));
    if (e != hipSuccess) return hipfail(e, "rvq_ema_step(assign)");
    e = mvq::launch_ema_update(z_tokens, idx, books, work, batch, dim, t, nb, k, decay, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "rvq_ema_step(update)");
}

int mvq_dac_rvq_f32(const float* z, const float* in_w, const float* in_b, const float* codebook,
                    const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                    int batch, int c, int t, int nq_use, int k, int dc, void* stream)
{
    return mvq_dac_rvq_items_f32(z, in_w, in_b, codebook, out_w, out_b, zq, codes, latents, nullptr, batch, c, t, nq_use, k, dc, stream);
}

int mvq_dac_rvq_items_f32(const float* z, const float* in_w, const float* in_b, const float* codebook,
                          const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                          const int32_t* nq_item, int batch, int c, int t, int nq_use, int k, int dc, void* stream)
{
    return mvq_dac_rvq_prepared_f32(z, in_w, in_b, codebook, nullptr, nullptr, out_w, out_b, zq, codes, latents, nq_item,
                                    batch, c, t, nq_use, k, dc, stream);
}

int mvq_dac_rvq_prepare_f32(const float* codebook, float* cb_normalised, float* cb_norm2, int nq, int k, int dc, void* stream)
{
    if (nq < 0 || k <= 0 || dc <= 0) return fail(MVQ_EINVAL, "dac_rvq_prepare: bad shape");
    if (nq == 0) return MVQ_OK;
    if (!codebook || !cb_normalised || !cb_norm2) return fail(MVQ_EINVAL, "dac_rvq_prepare: null tensor");
    hipError_t e = mvq::launch_dac_rvq_prepare(codebook, cb_normalised, cb_norm2, nq, k, dc, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "dac_rvq_prepare");
}

int mvq_dac_rvq_prepared_f32(const float* z, const float* in_w, const float* in_b, const float* codebook,
                             const float* cb_normalised, const float* cb_norm2,
                             const float* out_w, const float* out_b, float* zq, int32_t* codes, float* latents,
                             const int32_t* nq_item, int batch, int c, int t, int nq_use, int k, int dc, void* stream)
{
    if ((cb_normalised != nullptr) != (cb_norm2 != nullptr)) return fail(MVQ_EINVAL, "dac_rvq: cb_normalised and cb_norm2 go together");
    if (batch < 0 || t < 0 || (c != 1024 && c != 512 && c != 256) || dc != 8 || (k * dc) % 4 != 0 || nq_use <= 0 || k <= 0 || dc <= 0 || dc > 16)
        return fail(MVQ_EINVAL, "dac_rvq: bad shape B=%d C=%d T=%d nq=%d K=%d Dc=%d (C in {256,512,1024}, Dc = 8)", batch, c, t, nq_use, k, dc);
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!z || !zq || !codes || !latents || !in_w || !in_b || !codebook || !out_w || !out_b)
        return fail(MVQ_EINVAL, "dac_rvq: null tensor");
    const size_t lds = ((size_t)k * dc + k + (size_t)dc * c + 16 * (size_t)dc * 16 + 2 * (size_t)dc * 16 + 2 * 16 * 16) * sizeof(float);
    if (lds > 160 * 1024) return fail(MVQ_EUNSUPPORTED, "dac_rvq: K*Dc too large for LDS (%zu bytes)", lds);
    hipError_t e = mvq::launch_dac_rvq(z, in_w, in_b, codebook, out_w, out_b, zq, codes, latents, nq_item, batch, c, t, nq_use, k, dc, S(stream),
                                       cb_normalised, cb_norm2);
    return e == hipSuccess ? MVQ_OK : hipfail(e, "dac_rvq");
}

int mvq_layernorm_c_f32(const float* x, const float* pe, const float* gamma, const float* beta, float* y,
                        int batch, int c, int t, size_t stride_b, size_t stride_c,
                        float eps, int do_tanh, float post_scale, void* stream)
{
    return mvq_layernorm_c_sub_f32(x, nullptr, pe, gamma, beta, y, batch, c, t, stride_b, stride_c, eps, do_tanh, post_scale, stream);
}

int mvq_layernorm_c_sub_f32(const float* x, const float* sub, const float* pe, const float* gamma, const float* beta, float* y,
                            int batch, int c, int t, size_t stride_b, size_t stride_c,
                            float eps, int do_tanh, float post_scale, void* stream)
{
    if (c <= 0 || batch < 0 || t < 0) return fail(MVQ_EINVAL, "layernorm_c: bad shape");
    if (batch == 0 || t == 0) return MVQ_OK;                      /* empty chunk (Tk == 0 at a file end) */
    if (!x || !gamma || !beta || !y) return fail(MVQ_EINVAL, "layernorm_c: null tensor");
    if (stride_b == 0 && stride_c == 0) { stride_b = (size_t)c * t; stride_c = (size_t)t; }
    hipError_t e = mvq::launch_layernorm_c(x, pe, gamma, beta, y, batch, c, t, stride_b, stride_c, eps, do_tanh, post_scale, sub, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "layernorm_c");
}

int mvq_attention_f32(const float* q, const float* k, const float* v, float* ctx,
                      int batch, int heads, int dh, int tq, int tk,
                      size_t q_stride_b, size_t q_stride_c, size_t k_stride_b, size_t k_stride_c, void* stream)
{
    if (batch < 0 || heads <= 0 || dh <= 0 || tq < 0 || tk < 0 || tk > 64 || tq > 64 || (size_t)dh * (tq + 2 * tk) + (size_t)tq * tk > 16384)
        return fail(MVQ_EINVAL, "attention: bad shape (Tq, Tk <= 64; head slices must fit 64 KiB of LDS)");
    if (batch == 0 || tq == 0) return MVQ_OK;
    if (!q || !ctx || (tk > 0 && (!k || !v))) return fail(MVQ_EINVAL, "attention: null tensor");
    const size_t c = (size_t)heads * dh;
    if (q_stride_b == 0 && q_stride_c == 0) { q_stride_b = c * tq; q_stride_c = (size_t)tq; }
    if (k_stride_b == 0 && k_stride_c == 0) { k_stride_b = c * tk; k_stride_c = (size_t)tk; }
    hipError_t e = mvq::launch_attention(q, k, v, ctx, batch, heads, dh, tq, tk, q_stride_b, q_stride_c, k_stride_b, k_stride_c, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "attention");
}

int mvq_gelu_f32(const float* x, float* y, size_t n, void* stream)
{
    if ((!x || !y) && n) return fail(MVQ_EINVAL, "gelu: null tensor");
    hipError_t e = mvq::launch_gelu(x, y, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "gelu");
}

int mvq_sub3d_f32(const float* a, size_t a_sb, size_t a_sc, const float* b, size_t b_sb, size_t b_sc,
                  float* y, size_t y_sb, size_t y_sc, int batch, int c, int n, void* stream)
{
    if ((!a || !b || !y) && batch && c && n) return fail(MVQ_EINVAL, "sub3d: null tensor");
    if (batch < 0 || c < 0 || n < 0) return fail(MVQ_EINVAL, "sub3d: bad shape");
    hipError_t e = mvq::launch_strided3d(a, a_sb, a_sc, b, b_sb, b_sc, y, y_sb, y_sc, batch, c, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "sub3d");
}

int mvq_copy3d_f32(const float* a, size_t a_sb, size_t a_sc, float* y, size_t y_sb, size_t y_sc,
                   int batch, int c, int n, void* stream)
{
    if ((!a || !y) && batch && c && n) return fail(MVQ_EINVAL, "copy3d: null tensor");
    if (batch < 0 || c < 0 || n < 0) return fail(MVQ_EINVAL, "copy3d: bad shape");
    hipError_t e = mvq::launch_strided3d(a, a_sb, a_sc, nullptr, 0, 0, y, y_sb, y_sc, batch, c, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "copy3d");
}

int mvq_align_xcorr_f32(const float* ref, const float* est, int t, int max_shift, float* corr, int32_t* scratch,
                        int32_t* best_shift, void* stream)
{
    if (t < 0 || max_shift < 0) return fail(MVQ_EINVAL, "align_xcorr: bad shape");
    if (!ref || !est || !corr || !scratch || !best_shift) return fail(MVQ_EINVAL, "align_xcorr: null tensor");
    hipError_t e = mvq::launch_align_xcorr(ref, est, t, max_shift, corr, scratch, best_shift, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "align_xcorr");
}

int mvq_align_xcorr_batch_f32(const float* ref, const float* est, int batch, int t, int max_shift, float* corr, int32_t* scratch,
                              int32_t* best_shift, void* stream)
{
    if (batch < 0 || t < 0 || max_shift < 0) return fail(MVQ_EINVAL, "align_xcorr_batch: bad shape");
    if (batch == 0) return MVQ_OK;
    if (!ref || !est || !corr || !scratch || !best_shift) return fail(MVQ_EINVAL, "align_xcorr_batch: null tensor");
    hipError_t e = mvq::launch_align_xcorr_batch(ref, est, batch, t, max_shift, corr, scratch, best_shift, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "align_xcorr_batch");
}

int mvq_resample_ragged_f32(const float* x, const float* kern, float* y, const int32_t* off, const int32_t* len, int32_t* len_out,
                            int batch, int pitch, int lout_pitch, int orig, int newf, int width, int ks, void* stream)
{
    if (batch < 0 || pitch < 0 || lout_pitch < 0 || orig <= 0 || newf <= 0 || width < 0 || ks != 2 * width + orig)
        return fail(MVQ_EINVAL, "resample_ragged: bad shape (ks must be 2*width + orig)");
    /* off / len live on the device and cannot be checked here: the kernel clamps every slice to its row (off >= 0,
     * off + len <= pitch) and writes at most lout_pitch samples per row; len_out reports the length actually produced */
    if (batch == 0 || lout_pitch == 0) return MVQ_OK;
    if (!x || !kern || !y || !off || !len) return fail(MVQ_EINVAL, "resample_ragged: null tensor");
    hipError_t e = mvq::launch_resample_ragged(x, kern, y, off, len, len_out, batch, pitch, lout_pitch, orig, newf, width, ks, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "resample_ragged");
}

/* ---- backward (input-gradient) entry points: SURVEY.md section 8f row f1 ------------------------------------- */

size_t mvq_conv1d_dgrad_packed_floats(int cin, int cout, int ks)
{
    if (cin <= 0 || cout <= 0 || ks <= 0) return 0;
    return (size_t)cout * ks * mvq::conv_mpad(cin);
}

int mvq_conv1d_pack_dgrad_f32(const float* w, float* wp, int cin, int cout, int ks, void* stream)
{
    if (!w || !wp || cin <= 0 || cout <= 0 || ks <= 0) return fail(MVQ_EINVAL, "conv1d_pack_dgrad: bad argument");
    hipError_t e = mvq::launch_pack_conv1d_dgrad(w, wp, cin, cout, ks, mvq::conv_mpad(cin), S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_pack_dgrad");
}

int mvq_conv_transpose1d_pack_dgrad_f32(const float* w, float* wp, int cin, int cout, int ks, void* stream)
{
    if (!w || !wp || cin <= 0 || cout <= 0 || ks <= 0) return fail(MVQ_EINVAL, "conv_transpose1d_pack_dgrad: bad argument");
    hipError_t e = mvq::launch_pack_convtr_dgrad(w, wp, cin, cout, ks, mvq::conv_mpad(cin), S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv_transpose1d_pack_dgrad");
}

int mvq_conv1d_dgrad_f32(const float* gy, const float* wp_dgrad, const float* dsnake_src, const float* dsnake_alpha,
                         const float* residual, float* gx,
                         int batch, int cin, int tin, int cout, int tout, int ks, int stride, int dil, int pad, void* stream)
{
    /* gradient w.r.t. the input of a forward conv x[B,cin,tin] -> y[B,cout,tout]:
     *   stride 1 : conv1d over gy with the flipped/transposed image, same dilation, padding (ks-1)*dil - pad
     *   (forward was a ConvTranspose1d with kernel ks, stride, pad; its input-gradient is a strided conv): stride > 1 */
    if (batch < 0 || cin <= 0 || cout <= 0 || tin < 0 || tout < 0 || ks <= 0 || stride <= 0 || dil <= 0 || pad < 0)
        return fail(MVQ_EINVAL, "conv1d_dgrad: bad shape");
    if (batch == 0 || tin == 0) return MVQ_OK;
    if (!gy || !wp_dgrad || !gx) return fail(MVQ_EINVAL, "conv1d_dgrad: null tensor");
    if ((dsnake_src != nullptr) != (dsnake_alpha != nullptr)) return fail(MVQ_EINVAL, "conv1d_dgrad: dsnake_src and dsnake_alpha go together");
    const int bpad = stride == 1 ? (ks - 1) * dil - pad : pad;
    if (bpad < 0) return fail(MVQ_EUNSUPPORTED, "conv1d_dgrad: padding larger than the receptive field");
    const int olen = conv_out_len(tout, ks, stride, dil, bpad);
    if (olen != tin) return fail(MVQ_EINVAL, "conv1d_dgrad: shapes inconsistent (got %d input-gradient samples, expected %d)", olen, tin);
    mvq::ConvArgs a{};
    a.x = gy; a.wp = wp_dgrad; a.residual = residual; a.y = gx;
    a.B = batch; a.Cin = cout; a.Tin = tout; a.Cout = cin; a.Tout = tin; a.pad = bpad; a.Mpad = mvq::conv_mpad(cin);
    a.Mrows = cin; a.Ncols = tin; a.up_s = 1; a.dsn_src = dsnake_src; a.dsn_alpha = dsnake_alpha;
    hipError_t e = dispatch_conv1d(a, ks, stride, dil, S(stream));
    if (e == hipErrorInvalidValue) {
        (void)hipGetLastError();
        mvq::DirectConvArgs d{gy, wp_dgrad, nullptr, nullptr, residual, nullptr, gx, batch, cout, tout, cin, tin, ks, stride, dil, bpad,
                              a.Mpad, 0, nullptr, nullptr, dsnake_src, dsnake_alpha};
        e = mvq::launch_conv1d_direct(d, S(stream));
    }
    return e == hipSuccess ? MVQ_OK : hipfail(e, "conv1d_dgrad");
}

/* backward of the reference-owned trainable modules (tolerance-checked against torch autograd) */
int mvq_layernorm_c_bwd_f32(const float* x, const float* pe, const float* gamma, const float* g, float* gx,
                            float* dgamma, float* dbeta, float* stats, int batch, int c, int t,
                            size_t stride_b, size_t stride_c, float eps, void* stream)
{
    if (c <= 0 || batch < 0 || t < 0) return fail(MVQ_EINVAL, "layernorm_c_bwd: bad shape");
    if (batch == 0 || t == 0) return MVQ_OK;
    if (!x || !gamma || !g || !dgamma || !dbeta || !stats) return fail(MVQ_EINVAL, "layernorm_c_bwd: null tensor");
    if (stride_b == 0 && stride_c == 0) { stride_b = (size_t)c * t; stride_c = (size_t)t; }
    hipError_t e = mvq::launch_layernorm_bwd(x, pe, gamma, g, gx, dgamma, dbeta, stats, batch, c, t, stride_b, stride_c, eps, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "layernorm_c_bwd");
}

int mvq_gelu_bwd_f32(const float* x, const float* g, float* gx, size_t n, void* stream)
{
    if ((!x || !g || !gx) && n) return fail(MVQ_EINVAL, "gelu_bwd: null tensor");
    hipError_t e = mvq::launch_gelu_bwd(x, g, gx, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "gelu_bwd");
}

int mvq_scale_tanh_f32(const float* u, float scale, float* y, size_t n, void* stream)
{
    if ((!u || !y) && n) return fail(MVQ_EINVAL, "scale_tanh: null tensor");
    hipError_t e = mvq::launch_scale_tanh(u, scale, y, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "scale_tanh");
}

int mvq_scale_tanh_bwd_f32(const float* u, const float* g, float scale, float* gu, float* partial, int n_partial, size_t n,
                           void* stream)
{
    if (((!u || !g || !gu) && n) || !partial || n_partial <= 0 || n_partial > 4096) return fail(MVQ_EINVAL, "scale_tanh_bwd: bad argument");
    hipError_t e = mvq::launch_scale_tanh_bwd(u, g, scale, gu, partial, n_partial, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "scale_tanh_bwd");
}

int mvq_attention_bwd_f32(const float* q, const float* k, const float* v, const float* g, float* gq, float* gk, float* gv,
                          int batch, int heads, int dh, int tq, int tk,
                          size_t q_stride_b, size_t q_stride_c, size_t k_stride_b, size_t k_stride_c, void* stream)
{
    if (batch < 0 || heads <= 0 || dh <= 0 || tq < 0 || tk < 0 || tq > 32 || tk > 32) return fail(MVQ_EINVAL, "attention_bwd: bad shape (Tq, Tk <= 32)");
    if (batch == 0 || tq == 0) return MVQ_OK;
    if (!q || !g || !gq || (tk > 0 && (!k || !v || !gk || !gv))) return fail(MVQ_EINVAL, "attention_bwd: null tensor");
    const size_t c = (size_t)heads * dh;
    if (q_stride_b == 0 && q_stride_c == 0) { q_stride_b = c * tq; q_stride_c = (size_t)tq; }
    if (k_stride_b == 0 && k_stride_c == 0) { k_stride_b = c * tk; k_stride_c = (size_t)tk; }
    hipError_t e = mvq::launch_attention_bwd(q, k, v, g, gq, gk, gv, batch, heads, dh, tq, tk, q_stride_b, q_stride_c, k_stride_b, k_stride_c, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "attention_bwd");
}

int mvq_mul_scaled_f32(const float* a, const float* b, float scale, float* out, size_t n, void* stream)
{
    if ((!a || !b || !out) && n) return fail(MVQ_EINVAL, "mul_scaled: null tensor");
    hipError_t e = mvq::launch_mul_scaled(a, b, scale, out, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "mul_scaled");
}

int mvq_transpose2d_f32(const float* in, float* out, int rows, int cols, void* stream)
{
    if (rows < 0 || cols < 0) return fail(MVQ_EINVAL, "transpose2d: bad shape");
    if ((!in || !out) && rows && cols) return fail(MVQ_EINVAL, "transpose2d: null tensor");
    hipError_t e = mvq::launch_transpose2d(in, out, rows, cols, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "transpose2d");
}

int mvq_rowsum_f32(const float* in, float* out, int rows, int cols, int accumulate, void* stream)
{
    if (rows < 0 || cols < 0) return fail(MVQ_EINVAL, "rowsum: bad shape");
    if ((!in || !out) && rows) return fail(MVQ_EINVAL, "rowsum: null tensor");
    hipError_t e = mvq::launch_rowsum(in, out, rows, cols, accumulate, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "rowsum");
}

/* training losses (row f2): the glue kernels around the STFT-as-GEMM; see include/mvq.h */
#define MVQ_LOSS_CALL(cond, what, call) do { if (cond) return fail(MVQ_EINVAL, what ": bad argument"); \
    hipError_t e_ = (call); return e_ == hipSuccess ? MVQ_OK : hipfail(e_, what); } while (0)

int mvq_stft_frames_f32(const float* x, const float* window, float* out, int batch, int t, int n_fft, int hop, int nframes,
                        size_t ncols, size_t col0, void* stream)
{
    MVQ_LOSS_CALL(!x || !window || !out || batch < 0 || t <= n_fft / 2 || n_fft <= 0 || hop <= 0 || nframes != 1 + t / hop ||
                  col0 + (size_t)batch * nframes > ncols, "stft_frames",
                  mvq::launch_stft_frames(x, window, out, batch, t, n_fft, hop, nframes, ncols, col0, S(stream)));
}
int mvq_spec_mag_f32(const float* spec, float* mag, int f, int fp, size_t ncols, float eps, void* stream)
{
    MVQ_LOSS_CALL(!spec || !mag || f <= 0 || fp < f, "spec_mag", mvq::launch_spec_mag(spec, mag, f, fp, ncols, eps, S(stream)));
}
int mvq_spec_loss_partial_f32(const float* mag, float* partial, int p, int f, int batch, int nframes, size_t ncols, void* stream)
{
    MVQ_LOSS_CALL(!mag || !partial || p <= 0 || f <= 0 || batch < 0 || (size_t)2 * batch * nframes > ncols, "spec_loss_partial",
                  mvq::launch_spec_loss_partial(mag, partial, p, f, batch, nframes, ncols, S(stream)));
}
int mvq_spec_grad_f32(const float* spec, const float* mag, const float* coef_a, float coef_b, const float* extra, float* g,
                      int f, int fp, int batch, int nframes, size_t ncols, float eps, void* stream)
{
    MVQ_LOSS_CALL(!spec || !mag || !g || f <= 0 || fp < f || batch < 0 || (size_t)2 * batch * nframes > ncols, "spec_grad",
                  mvq::launch_spec_grad(spec, mag, coef_a, coef_b, extra, g, f, fp, batch, nframes, ncols, eps, S(stream)));
}
int mvq_overlap_add_f32(const float* dframes, const float* window, float* dy, int batch, int t, int n_fft, int hop, int nframes,
                        void* stream)
{
    MVQ_LOSS_CALL(!dframes || !window || !dy || batch < 0 || t <= n_fft / 2 || hop <= 0 || nframes != 1 + t / hop, "overlap_add",
                  mvq::launch_overlap_add(dframes, window, dy, batch, t, n_fft, hop, nframes, S(stream)));
}
int mvq_l1_loss_f32(const float* y, const float* tgt, float* partial, int p, float* dy, float coef, size_t n, void* stream)
{
    MVQ_LOSS_CALL(!y || !tgt || !partial || p <= 0 || p > 4096, "l1_loss", mvq::launch_l1_loss(y, tgt, partial, p, dy, coef, n, S(stream)));
}
int mvq_mel_max_f32(const float* mel, float* maxv, int* argmax, int n_mels, int batch, int nframes, size_t ncols, void* stream)
{
    MVQ_LOSS_CALL(!mel || !maxv || !argmax || n_mels <= 0 || batch < 0 || (size_t)2 * batch * nframes > ncols, "mel_max",
                  mvq::launch_mel_max(mel, maxv, argmax, n_mels, batch, nframes, ncols, S(stream)));
}
int mvq_mel_cos_f32(const float* mel, const float* maxv, float* cosv, float* dmel, float* dden, float coef, int n_mels, int batch,
                    int nframes, size_t ncols, float eps, int use_log, void* stream)
{
    MVQ_LOSS_CALL(!mel || !maxv || !cosv || (dmel && (!dden || !use_log)) || n_mels <= 0 || batch < 0 || (size_t)2 * batch * nframes > ncols, "mel_cos",
                  mvq::launch_mel_cos(mel, maxv, cosv, dmel, dden, coef, n_mels, batch, nframes, ncols, eps, use_log, S(stream)));
}
int mvq_mel_max_grad_f32(const float* dden, const float* maxv, const int* argmax, float* dmel, int batch, int nframes, float eps,
                         void* stream)
{
    MVQ_LOSS_CALL(!dden || !maxv || !argmax || !dmel || batch < 0, "mel_max_grad",
                  mvq::launch_mel_max_grad(dden, maxv, argmax, dmel, batch, nframes, eps, S(stream)));
}

int mvq_resample_f32(const float* x, const float* kern, float* y, int batch, int len, int len_out, int orig, int newf,
                     int width, int ks, void* stream)
{
    if (batch < 0 || len < 0 || len_out < 0 || orig <= 0 || newf <= 0 || width < 0 || ks != 2 * width + orig)
        return fail(MVQ_EINVAL, "resample: bad shape (ks must be 2*width + orig)");
    if ((long long)len_out > ((long long)newf * len + orig - 1) / orig) return fail(MVQ_EINVAL, "resample: len_out exceeds ceil(new*len/orig)");
    if (batch == 0 || len_out == 0) return MVQ_OK;
    if (!x || !kern || !y) return fail(MVQ_EINVAL, "resample: null tensor");
    hipError_t e = mvq::launch_resample(x, kern, y, batch, len, len_out, orig, newf, width, ks, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "resample");
}

int mvq_sumsq_partial_f32(const float* x, float* partial, int n_partial, size_t n, void* stream)
{
    if (!partial || n_partial <= 0 || n_partial > 4096 || (!x && n)) return fail(MVQ_EINVAL, "sumsq_partial: bad argument");
    hipError_t e = mvq::launch_sumsq_partial(x, partial, n_partial, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "sumsq_partial");
}

int mvq_adamw_f32(float* p, const float* g, float* m, float* v, const float* clip_coef, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream)
{
    if (step < 1 || beta1 < 0 || beta1 >= 1 || beta2 < 0 || beta2 >= 1) return fail(MVQ_EINVAL, "adamw: bad hyper-parameter");
    if ((!p || !g || !m || !v) && n) return fail(MVQ_EINVAL, "adamw: null tensor");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipError_t e = mvq::launch_adamw(p, g, m, v, clip_coef, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "adamw");
}

int mvq_mul_dtanh_f32(const float* g, const float* y, float* out, size_t n, void* stream)
{
    if ((!g || !y || !out) && n) return fail(MVQ_EINVAL, "mul_dtanh: null tensor");
    hipError_t e = mvq::launch_mul_dtanh(g, y, out, n, S(stream));
    return e == hipSuccess ? MVQ_OK : hipfail(e, "mul_dtanh");
}

}  // extern "C"
