/* The C ABI used from plain C -- no Python, no torch: device buffers from the HIP runtime, entry points of
 * include/elliot_hip.h.  What a maintainer of a C / Go / Java host would bind (cgo / JNI over exactly these calls).
 *
 *   build:  gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *               -Lelliot_amd/csrc -lelliot_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/elliot_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm \
 *               -o /tmp/c_abi_demo          (-D__HIP_PLATFORM_AMD__ is what <hip/hip_runtime_api.h> wants from a non-hipcc compiler)
 *   run:    /tmp/c_abi_demo            (needs an MI355X)
 *
 * It trains a tiny BPR-MF model for a few steps (el_bpr_sample + el_bprmf_train_step, TF-dense Adam), asks for the top-5 items
 * of every user with the training items masked (el_score_topk) and checks the lists against a scalar recomputation on the host.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "elliot_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_EL(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, el_last_error()); return 1; } } while (0)

enum { U = 200, I = 300, F = 16, K = 5, B = 4096, STEPS = 5 };

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }

static void* dev_copy(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    if (host) (void)hipMemcpy(d, host, bytes, hipMemcpyHostToDevice); else (void)hipMemset(d, 0, bytes);
    return d;
}

int main(void) {
    /* train positives: user u likes items (7u + 3t) mod I, t < 6  -> CSR with ascending column indices */
    static int64_t indptr[U + 1];
    static int32_t indices[U * 6];
    for (int u = 0; u < U; ++u) {
        int32_t row[6];
        for (int t = 0; t < 6; ++t) row[t] = (7 * u + 3 * t) % I;
        for (int a = 0; a < 6; ++a) for (int b = a + 1; b < 6; ++b) if (row[b] < row[a]) { int32_t x = row[a]; row[a] = row[b]; row[b] = x; }
        indptr[u] = 6 * u;
        memcpy(indices + 6 * u, row, sizeof(row));
    }
    indptr[U] = 6 * U;
    static float Gu[U * F], Gi[I * F], Bi[I];
    unsigned seed = 1;
    for (int x = 0; x < U * F; ++x) Gu[x] = frand(&seed);
    for (int x = 0; x < I * F; ++x) Gi[x] = frand(&seed);

    el_ctx* ctx = NULL;
    CHECK_EL(el_ctx_create(0, &ctx));
    char name[64]; int cus = 0; int64_t hbm = 0;
    CHECK_EL(el_device_info(ctx, name, sizeof(name), &cus, &hbm));
    printf("device %s, %d CUs, %.0f GB; ABI %d\n", name, cus, hbm / 1e9, el_abi_version());

    el_bprmf_state st;
    memset(&st, 0, sizeof(st));
    st.U = U, st.I = I, st.F = F;
    st.Gu = dev_copy(Gu, sizeof(Gu)), st.Gi = dev_copy(Gi, sizeof(Gi)), st.Bi = dev_copy(Bi, sizeof(Bi));
    st.gGu = dev_copy(NULL, sizeof(Gu)), st.gGi = dev_copy(NULL, sizeof(Gi)), st.gBi = dev_copy(NULL, sizeof(Bi));
    st.mGu = dev_copy(NULL, sizeof(Gu)), st.vGu = dev_copy(NULL, sizeof(Gu));
    st.mGi = dev_copy(NULL, sizeof(Gi)), st.vGi = dev_copy(NULL, sizeof(Gi));
    st.mBi = dev_copy(NULL, sizeof(Bi)), st.vBi = dev_copy(NULL, sizeof(Bi));
    int64_t* d_indptr = dev_copy(indptr, sizeof(indptr));
    int32_t* d_indices = dev_copy(indices, sizeof(indices));
    int32_t* trip = dev_copy(NULL, 3 * B * sizeof(int32_t));
    double* d_loss = dev_copy(NULL, sizeof(double));
    size_t ws_bytes = el_bprmf_ws_bytes(B, U, I, F);
    void* ws = dev_copy(NULL, ws_bytes);

    const float lr = 0.01f;
    for (int t = 1; t <= STEPS; ++t) {
        CHECK_EL(el_bpr_sample(ctx, NULL, d_indptr, d_indices, U, I, 0, I, 42, (uint64_t)(t - 1) * B, B, trip, trip + B, trip + 2 * B));
        const float lr_t = lr * sqrtf(1.0f - powf(0.999f, (float)t)) / (1.0f - powf(0.9f, (float)t));
        CHECK_EL(el_bprmf_train_step(ctx, NULL, &st, trip, trip + B, trip + 2 * B, B, lr, 0.01f, 0.001f, EL_OPT_ADAM_TF_DENSE, t, lr_t,
                                     d_loss, EL_BPR_AUTO, ws, ws_bytes));
        double loss = 0;
        CHECK_HIP(hipMemcpy(&loss, d_loss, sizeof(loss), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemset(d_loss, 0, sizeof(double)));
        printf("step %d  loss/triplet %.5f\n", t, loss / B);
    }

    int32_t* d_idx = dev_copy(NULL, U * K * sizeof(int32_t));
    float* d_val = dev_copy(NULL, U * K * sizeof(float));
    size_t tk_bytes = el_score_topk_ws_bytes(U, I, F, K, indptr[U], EL_TOPK_AUTO);
    void* tk_ws = tk_bytes ? dev_copy(NULL, tk_bytes) : NULL;
    CHECK_EL(el_score_topk(ctx, NULL, st.Gu, st.Gi, st.Bi, 0, U, 0, I, F, d_indptr, d_indices, NULL, NULL, K, d_idx, d_val, EL_TOPK_AUTO,
                           tk_ws, tk_bytes));
    static int32_t idx[U * K];
    static float val[U * K];
    CHECK_HIP(hipMemcpy(idx, d_idx, sizeof(idx), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(val, d_val, sizeof(val), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Gu, st.Gu, sizeof(Gu), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Gi, st.Gi, sizeof(Gi), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Bi, st.Bi, sizeof(Bi), hipMemcpyDeviceToHost));

    /* host check: the same fp32 fma chain (bias + sum in k order), mask, (score desc, index asc) */
    int bad = 0;
    for (int u = 0; u < U; ++u) {
        float sc[I];
        for (int i = 0; i < I; ++i) {
            float a = 0.f;
            for (int f = 0; f < F; ++f) a = fmaf(Gu[u * F + f], Gi[i * F + f], a);
            sc[i] = (a + Bi[i]) + 0.0f;
        }
        for (int64_t e = indptr[u]; e < indptr[u + 1]; ++e) sc[indices[e]] = -INFINITY;
        for (int r = 0; r < K; ++r) {
            int best = -1;
            for (int i = 0; i < I; ++i) if (best < 0 || sc[i] > sc[best]) best = i;
            if (idx[u * K + r] != best || val[u * K + r] != sc[best]) ++bad;
            sc[best] = -INFINITY;
        }
    }
    printf("top-%d of %d users: %s (%d mismatching entries)\n", K, U, bad ? "MISMATCH" : "identical to the host recomputation", bad);
    CHECK_EL(el_ctx_destroy(ctx));
    return bad ? 2 : 0;
}
