/* The sharded paths from a plain C host: one process per GPU, RCCL through the library's own entry points (el_comm_*), no
 * Python, no torch, no MPI.  What SURVEY 8b asks of the boundary: "a C host can run the multi-GPU path".
 *
 *   build:  gcc -std=c99 -D_POSIX_C_SOURCE=200809L -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_multigpu.c \
 *               -Lelliot_amd/csrc -lelliot_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/elliot_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm \
 *               -o /tmp/c_abi_multigpu
 *   run:    /tmp/c_abi_multigpu [world]      (default: one rank per visible GPU; HSA_ENABLE_IPC_MODE_LEGACY=0 as for any RCCL job)
 *
 * The parent forks `world` ranks BEFORE anything touches HIP.  Rank 0 obtains the 128-byte communicator id
 * (el_comm_unique_id) and publishes it through a file (any out-of-band channel does); every rank then
 *   - all-reduces a gradient-like buffer              (el_allreduce_rows: the item gradients of the user-sharded BPR step)
 *   - reduce-scatters + all-gathers a table            (el_reduce_scatter_rows / el_allgather_rows: the dense user-gradient exchange
 *                                                       of the item-sharded step)
 *   - scores its ITEM SHARD for a block of users and all-gathers + merges the partial top-k lists
 *                                                      (el_score_topk on I/G items, el_allgather_topk, el_topk_merge)
 * and rank 0 checks the merged lists against the single-shard answer computed on its own GPU.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "elliot_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_EL(x) do { if ((x) != 0) { fprintf(stderr, "[rank %d] %s: %s\n", rank, #x, el_last_error()); return 1; } } while (0)

enum { U = 256, I = 4096, F = 32, K = 10, N = 1 << 16 };

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }

static int run_rank(int rank, int world, const char* id_path) {
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    const int dev = rank % (ndev > 0 ? ndev : 1);
    CHECK_HIP(hipSetDevice(dev));
    el_ctx* ctx = NULL;
    CHECK_EL(el_ctx_create(dev, &ctx));

    /* ---- communicator: rank 0 creates the id, the others read it ---------------------------------------------------- */
    unsigned char id[128];
    if (rank == 0) {
        CHECK_EL(el_comm_unique_id(id));
        char tmp[512];
        snprintf(tmp, sizeof(tmp), "%s.tmp", id_path);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "cannot publish the communicator id\n"); return 1; }
        fclose(f);
        rename(tmp, id_path);
    } else {
        FILE* f = NULL;
        const struct timespec tenth = {0, 100000000L};
        for (int tries = 0; tries < 600 && !(f = fopen(id_path, "rb")); ++tries) nanosleep(&tenth, NULL);
        if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "[rank %d] no communicator id\n", rank); return 1; }
        fclose(f);
    }
    el_comm* comm = NULL;
    CHECK_EL(el_comm_init(ctx, id, rank, world, &comm));
    int r2 = -1, w2 = -1;
    CHECK_EL(el_comm_rank(comm, &r2, &w2));
    if (r2 != rank || w2 != world) { fprintf(stderr, "[rank %d] communicator reports %d of %d\n", rank, r2, w2); return 1; }

    /* ---- all-reduce -------------------------------------------------------------------------------------------------- */
    float* h = (float*)malloc(N * sizeof(float));
    for (int x = 0; x < N; ++x) h[x] = (float)(rank + 1) * (float)(x % 7);
    float* d = NULL;
    CHECK_HIP(hipMalloc((void**)&d, N * sizeof(float)));
    CHECK_HIP(hipMemcpy(d, h, N * sizeof(float), hipMemcpyHostToDevice));
    CHECK_EL(el_allreduce_rows(ctx, comm, NULL, d, N));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h, d, N * sizeof(float), hipMemcpyDeviceToHost));
    const float tri = (float)(world * (world + 1) / 2);
    for (int x = 0; x < N; ++x) if (h[x] != tri * (float)(x % 7)) { fprintf(stderr, "[rank %d] all-reduce: element %d is %g\n", rank, x, h[x]); return 1; }

    /* ---- reduce-scatter + all-gather ------------------------------------------------------------------------------------ */
    const int64_t per = N / world;
    float *full = NULL, *own = NULL;
    CHECK_HIP(hipMalloc((void**)&full, per * world * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&own, per * sizeof(float)));
    for (int64_t x = 0; x < per * world; ++x) h[x] = (float)(rank + 1);
    CHECK_HIP(hipMemcpy(full, h, per * world * sizeof(float), hipMemcpyHostToDevice));
    CHECK_EL(el_reduce_scatter_rows(ctx, comm, NULL, full, own, per));
    CHECK_EL(el_allgather_rows(ctx, comm, NULL, own, full, per * (int64_t)sizeof(float)));
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(h, full, per * world * sizeof(float), hipMemcpyDeviceToHost));
    for (int64_t x = 0; x < per * world; ++x) if (h[x] != tri) { fprintf(stderr, "[rank %d] reduce-scatter/all-gather: %g\n", rank, h[x]); return 1; }

    /* ---- item-sharded top-k: partial lists per shard, all-gather, merge ------------------------------------------------- */
    static float Gu[U * F], Gi[I * F], Bi[I];
    unsigned seed = 7;                                   /* the same tables on every rank */
    for (int x = 0; x < U * F; ++x) Gu[x] = frand(&seed);
    for (int x = 0; x < I * F; ++x) Gi[x] = frand(&seed);
    for (int x = 0; x < I; ++x) Bi[x] = frand(&seed) * 0.1f;
    const int64_t lo = (int64_t)I * rank / world, hi = (int64_t)I * (rank + 1) / world;
    float *dGu = NULL, *dGi = NULL, *dBi = NULL, *pv = NULL, *av = NULL, *mv = NULL;
    int32_t *pi = NULL, *ai = NULL, *mi = NULL;
    CHECK_HIP(hipMalloc((void**)&dGu, sizeof(Gu)));
    CHECK_HIP(hipMalloc((void**)&dGi, sizeof(Gi)));
    CHECK_HIP(hipMalloc((void**)&dBi, sizeof(Bi)));
    CHECK_HIP(hipMemcpy(dGu, Gu, sizeof(Gu), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dGi, Gi, sizeof(Gi), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dBi, Bi, sizeof(Bi), hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc((void**)&pi, U * K * 4));
    CHECK_HIP(hipMalloc((void**)&pv, U * K * 4));
    CHECK_HIP(hipMalloc((void**)&ai, (size_t)world * U * K * 4));
    CHECK_HIP(hipMalloc((void**)&av, (size_t)world * U * K * 4));
    CHECK_HIP(hipMalloc((void**)&mi, U * K * 4));
    CHECK_HIP(hipMalloc((void**)&mv, U * K * 4));
    size_t wsb = el_score_topk_ws_bytes(U, I, F, K, 0, EL_TOPK_AUTO);
    void* ws = NULL;
    if (wsb) CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_EL(el_score_topk(ctx, NULL, dGu, dGi + lo * F, dBi + lo, 0, U, lo, hi - lo, F, NULL, NULL, NULL, NULL, K, pi, pv, EL_TOPK_AUTO, ws, wsb));
    CHECK_EL(el_allgather_topk(ctx, comm, NULL, pi, pv, U, K, ai, av));
    CHECK_EL(el_topk_merge(ctx, NULL, ai, av, world, U, K, mi, mv));
    CHECK_HIP(hipDeviceSynchronize());
    if (rank == 0) {
        int32_t *ri = NULL, *hm = (int32_t*)malloc(U * K * 4), *hr = (int32_t*)malloc(U * K * 4);
        float *rv = NULL, *hmv = (float*)malloc(U * K * 4), *hrv = (float*)malloc(U * K * 4);
        CHECK_HIP(hipMalloc((void**)&ri, U * K * 4));
        CHECK_HIP(hipMalloc((void**)&rv, U * K * 4));
        CHECK_EL(el_score_topk(ctx, NULL, dGu, dGi, dBi, 0, U, 0, I, F, NULL, NULL, NULL, NULL, K, ri, rv, EL_TOPK_AUTO, ws, wsb));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(hm, mi, U * K * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hr, ri, U * K * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hmv, mv, U * K * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hrv, rv, U * K * 4, hipMemcpyDeviceToHost));
        if (memcmp(hm, hr, U * K * 4) != 0 || memcmp(hmv, hrv, U * K * 4) != 0) { fprintf(stderr, "merged item-shard lists differ from the single-shard lists\n"); return 1; }
        printf("world %d: all-reduce, reduce-scatter + all-gather exact; item-sharded top-%d of %d users over %d shards == single shard (indices and score bits)\n",
               world, K, U, world);
        fflush(stdout);                                  /* the rank leaves through _exit */
    }
    CHECK_EL(el_comm_destroy(comm));
    CHECK_EL(el_ctx_destroy(ctx));
    return 0;
}

int main(int argc, char** argv) {
    int world = argc > 1 ? atoi(argv[1]) : 0;
    if (world <= 0) {                                    /* default: one rank per visible GPU (asked in a child: no HIP in the parent) */
        int fd[2];
        if (pipe(fd) != 0) return 1;
        pid_t p = fork();
        if (p == 0) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) n = 0; if (write(fd[1], &n, sizeof(n)) != sizeof(n)) _exit(1); _exit(0); }
        int n = 0, stw = 0;
        if (read(fd[0], &n, sizeof(n)) != sizeof(n)) n = 0;
        waitpid(p, &stw, 0);
        world = n > 0 ? n : 1;
    }
    char id_path[256];
    snprintf(id_path, sizeof(id_path), "/tmp/el_comm_id_%d", (int)getpid());
    unlink(id_path);
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    pid_t* kids = (pid_t*)malloc(sizeof(pid_t) * world);
    for (int r = 0; r < world; ++r) {
        kids[r] = fork();
        if (kids[r] == 0) _exit(run_rank(r, world, id_path));
    }
    int bad = 0;
    for (int r = 0; r < world; ++r) {
        int st = 0;
        waitpid(kids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
    }
    unlink(id_path);
    if (bad) { fprintf(stderr, "a rank failed\n"); return 1; }
    return 0;
}
