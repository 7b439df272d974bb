// The drop-in single-call symbols from N native threads (no interpreter lock in the way): calls per second of the whole process.
//   tools/probes/single_call_mt <path to libsprintz_mi355x.so> [threads ...]        (default 1 8 16 64)
// Every thread alternates sprintz_mi355x_decompress_xff_16b / sprintz_mi355x_compress_xff_16b on its own 10 KB chunk (uint16 x 8 columns,
// a random walk) and checks the round trip at the end.
#include <dlfcn.h>
#include <pthread.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int64_t (*comp_fn)(const uint16_t*, uint32_t, int16_t*, uint16_t, int);
typedef int64_t (*decomp_fn)(const int16_t*, uint16_t*);
static comp_fn g_comp;
static decomp_fn g_decomp;
static pthread_barrier_t g_go;
static const uint32_t kLen = 5120, kD = 8;
struct Job { int calls; std::vector<uint16_t> raw, back; std::vector<int16_t> comp; bool ok; };

static void* work(void* p)
{
    Job* j = (Job*)p;
    g_comp(j->raw.data(), kLen, j->comp.data(), kD, 1);            // first call of the thread: scratch, stream
    g_decomp(j->comp.data(), j->back.data());
    pthread_barrier_wait(&g_go);
    for (int i = 0; i < j->calls; i++) {
        g_decomp(j->comp.data(), j->back.data());
        g_comp(j->raw.data(), kLen, j->comp.data(), kD, 1);
    }
    j->ok = memcmp(j->raw.data(), j->back.data(), kLen * 2) == 0;
    return nullptr;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s libsprintz_mi355x.so [threads ...]\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    g_comp = (comp_fn)dlsym(h, "sprintz_mi355x_compress_xff_16b");
    g_decomp = (decomp_fn)dlsym(h, "sprintz_mi355x_decompress_xff_16b");
    if (!g_comp || !g_decomp) { fprintf(stderr, "symbols missing\n"); return 2; }
    std::vector<int> counts;
    for (int i = 2; i < argc; i++) counts.push_back(atoi(argv[i]));
    if (counts.empty()) counts = {1, 8, 16, 64};
    for (int nt : counts) {
        std::vector<Job> jobs(nt);
        uint64_t s = 88172645463325252ull;
        for (auto& j : jobs) {
            j.calls = getenv("CALLS") ? atoi(getenv("CALLS")) : (nt >= 32 ? 400 : 1500);      // CALLS=1000000: a soak run (watch VmRSS)
            j.raw.resize(kLen + 64); j.back.assign(kLen + 64, 0); j.comp.assign(kLen * 3 / 2 + 64, 0);
            uint16_t v[kD] = {0};
            for (uint32_t r = 0; r < kLen / kD; r++)
                for (uint32_t d = 0; d < kD; d++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v[d] = (uint16_t)(v[d] + (int)(s % 17) - 8); j.raw[r * kD + d] = v[d]; }
        }
        pthread_barrier_init(&g_go, nullptr, nt + 1);
        std::vector<pthread_t> th(nt);
        for (int i = 0; i < nt; i++) pthread_create(&th[i], nullptr, work, &jobs[i]);
        pthread_barrier_wait(&g_go);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nt; i++) pthread_join(th[i], nullptr);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        bool ok = true;
        for (auto& j : jobs) ok = ok && j.ok;
        printf("%3d threads: %9.0f calls/s  (%.1f us per call per thread)%s\n", nt, 2.0 * nt * jobs[0].calls / dt, dt / (2.0 * jobs[0].calls) * 1e6, ok ? "" : "  ROUND TRIP FAILED");
        pthread_barrier_destroy(&g_go);
        if (FILE* f = fopen("/proc/self/status", "r")) {             // resident set after the pass: a runtime that never retired the launches would show here
            char line[256];
            while (fgets(line, sizeof line, f)) if (!strncmp(line, "VmRSS", 5)) printf("      %s", line);
            fclose(f);
        }
    }
    return 0;
}
