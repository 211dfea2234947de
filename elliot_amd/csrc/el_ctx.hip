// Context + error plumbing of the C ABI (include/elliot_hip.h).
#include <stdarg.h>
#include <stdlib.h>
#include "el_common.h"

static thread_local char g_el_err[1024] = "";
thread_local el_ctx* g_el_cur_ctx = nullptr;

void el_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_el_err, sizeof(g_el_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* el_last_error(void) { return g_el_err; }

extern "C" int el_abi_version(void) { return EL_ABI_VERSION; }

// ---- options ----------------------------------------------------------------------------------------------------------------
struct el_opt_entry {
    const char* name;
    double el_options::*field;
};
static const el_opt_entry g_el_opts[] = {
    {"ichunk", &el_options::ichunk}, {"uchunk", &el_options::uchunk}, {"loop_graph", &el_options::loop_graph},
    {"gemm_split", &el_options::gemm_split}, {"gemm_xcd", &el_options::gemm_xcd}, {"nmf_side", &el_options::nmf_side}, {"nmf_head4", &el_options::nmf_head4},
    {"vae_side", &el_options::vae_side}, {"nmf_screen_maxfrac", &el_options::nmf_screen_maxfrac},
    {"screen_stride", &el_options::screen_stride}, {"screen_ka", &el_options::screen_ka},
    {"screen_prof", &el_options::screen_prof},
};

// EL_<NAME> in the environment gives an option its initial value (the one place the library reads the environment)
static void el_options_from_env(el_options* o) {
    for (const auto& e : g_el_opts) {
        char env[64] = "EL_";
        size_t k = 3;
        for (const char* c = e.name; *c && k + 1 < sizeof(env); ++c) env[k++] = (char)((*c >= 'a' && *c <= 'z') ? *c - 32 : *c);
        env[k] = 0;
        if (const char* v = getenv(env)) o->*(e.field) = atof(v);
    }
}

extern "C" int el_ctx_set_option(el_ctx* ctx, const char* name, double value) {
    EL_REQUIRE(ctx != nullptr && name != nullptr, "el_ctx_set_option: null argument");
    for (const auto& e : g_el_opts)
        if (strcmp(e.name, name) == 0) {
            ctx->opt.*(e.field) = value;
            return 0;
        }
    el_set_error("el_ctx_set_option: unknown option '%s'", name);
    return 2;
}

extern "C" int el_ctx_get_option(el_ctx* ctx, const char* name, double* value) {
    EL_REQUIRE(ctx != nullptr && name != nullptr && value != nullptr, "el_ctx_get_option: null argument");
    for (const auto& e : g_el_opts)
        if (strcmp(e.name, name) == 0) {
            *value = ctx->opt.*(e.field);
            return 0;
        }
    el_set_error("el_ctx_get_option: unknown option '%s'", name);
    return 2;
}

extern "C" int el_ctx_create(int device, el_ctx** out) {
    EL_REQUIRE(out != nullptr, "el_ctx_create: out is NULL");
    int n = 0;
    EL_CHECK_HIP(hipGetDeviceCount(&n));
    EL_REQUIRE(device >= 0 && device < n, "el_ctx_create: device %d out of range (%d visible)", device, n);
    hipDeviceProp_t prop;
    EL_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    // This library carries gfx950 code objects only: fail loudly anywhere else.
    EL_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
               "el_ctx_create: device %d is %s; libelliot_hip.so is built for gfx950 (MI355X) only", device,
               prop.gcnArchName);
    el_ctx* c = new el_ctx();
    c->device = device;
    c->cus = prop.multiProcessorCount;
    c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    strncpy(c->arch, prop.gcnArchName, sizeof(c->arch) - 1);
    c->arch[sizeof(c->arch) - 1] = 0;
    c->timing = false;
    el_options_from_env(&c->opt);
    EL_CHECK_HIP(hipSetDevice(device));
    if (hipMalloc((void**)&c->zeros, 256) != hipSuccess || hipMemset(c->zeros, 0, 256) != hipSuccess) {
        el_set_error("el_ctx_create: cannot allocate the context's scratch on device %d", device);
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

bool el_side_stream_ready(el_ctx* ctx) {
    if (!ctx) return false;
    if (!ctx->side && hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess) {
        ctx->side = nullptr;
        return false;
    }
    for (auto& e : ctx->side_ev)
        if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            e = nullptr;
            return false;
        }
    return true;
}

extern "C" int el_ctx_destroy(el_ctx* ctx) {
    if (!ctx) return 0;
    if (g_el_cur_ctx == ctx) g_el_cur_ctx = nullptr;
    for (auto& r : ctx->pending) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->pool) (void)hipEventDestroy(e);
    if (ctx->loop_graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)ctx->loop_graph_exec);
    for (auto e : ctx->side_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->zeros) (void)hipFree(ctx->zeros);
    if (ctx->lr_copied) {
        (void)hipEventSynchronize(ctx->lr_copied);
        (void)hipEventDestroy(ctx->lr_copied);
    }
    if (ctx->lr_pinned) (void)hipHostFree(ctx->lr_pinned);
    delete ctx;
    return 0;
}

extern "C" int el_timing_enable(el_ctx* ctx, int on) {
    EL_REQUIRE(ctx != nullptr, "el_timing_enable: null ctx");
    ctx->timing = on != 0;
    return 0;
}

extern "C" int el_timing_filter(el_ctx* ctx, const char* kernel_name) {
    EL_REQUIRE(ctx != nullptr, "el_timing_filter: null ctx");
    ctx->timing_only = kernel_name ? kernel_name : "";
    return 0;
}

extern "C" int el_tuning_mode(el_ctx* ctx, int on) {
    EL_REQUIRE(ctx != nullptr, "el_tuning_mode: null ctx");
    ctx->tuning = on != 0;
    return 0;
}

// Synchronises the recorded events and writes "name count total_ms\n" lines (aggregated per kernel
// name) into buf; clears the records.
extern "C" int el_timing_report(el_ctx* ctx, char* buf, int len) {
    EL_REQUIRE(ctx != nullptr && buf != nullptr && len > 0, "el_timing_report: bad arguments");
    if (int rc = el_bind(ctx)) return rc;
    struct Agg {
        const char* name;
        long count;
        double ms;
    };
    std::vector<Agg> agg;
    for (auto& r : ctx->pending) {
        EL_CHECK_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        EL_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        bool found = false;
        for (auto& a : agg)
            if (strcmp(a.name, r.name) == 0) {
                a.count++;
                a.ms += ms;
                found = true;
                break;
            }
        if (!found) agg.push_back({r.name, 1, (double)ms});
        ctx->pool.push_back(r.a);
        ctx->pool.push_back(r.b);
    }
    ctx->pending.clear();
    int off = 0;
    buf[0] = 0;
    for (auto& a : agg) {
        int n = snprintf(buf + off, (size_t)(len - off), "%s %ld %.6f\n", a.name, a.count, a.ms);
        if (n < 0 || n >= len - off) break;
        off += n;
    }
    return 0;
}

extern "C" int el_device_info(el_ctx* ctx, char* name, int len, int* cus, int64_t* hbm_bytes) {
    EL_REQUIRE(ctx != nullptr, "el_device_info: null ctx");
    if (name && len > 0) {
        strncpy(name, ctx->arch, (size_t)len - 1);
        name[len - 1] = 0;
    }
    if (cus) *cus = ctx->cus;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return 0;
}
