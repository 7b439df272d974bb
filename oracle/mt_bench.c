/*
 * mt_bench.c -- the CPU baseline of bench.py on MANY host cores.  TEST/BENCH INFRASTRUCTURE ONLY
 * (lives in liboracle.so; nothing under sprintz_amd/ links or calls it).
 *
 * Methodology: communicate/ubicomp/results.tex:4-6 times the reference on one thread; SURVEY.md 8(d)
 * asks for "1 thread and all physical cores (one chunk range per std::thread, no sharing)".  Round 2 did
 * the many-thread leg with Python threads around ctypes calls of ~80 us each: the threads convoyed on the
 * interpreter lock and 256 of them scaled 4x.  Here the threads are pthreads that never leave C:
 *
 *   - thread t owns the contiguous chunk range [n*t/T, n*(t+1)/T) and its own slice of the output
 *     (the reference's decoder stores whole 32-byte vectors, sprintz_xff_rle.cpp:1054,1116, i.e. up to 31
 *     bytes past a chunk: slices are `out_gap` bytes apart so that no thread writes into another's range);
 *   - optionally pinned to the logical CPU the caller names (one per physical core);
 *   - every thread first copies ITS slice of the compressed input into memory it allocates itself (NUMA-local by first
 *     touch), then all threads wait on one flag and make `reps` passes over their range;
 *   - the result is (latest end - earliest start) / reps: the sustained all-core time of ONE pass.
 *
 * The decoders themselves are handed in as function pointers (the compiled reference's
 * ref_decompress_chunks, or oracle_decompress_chunks; libzstd's HUF_decompress for the cfg4 chain), so
 * this file links against none of them.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <time.h>

typedef uint64_t (*chunks_fn)(int codec, int elem_bytes, const uint8_t* comp, const uint64_t* offsets, uint64_t nchunks,
                              uint32_t chunk_len, void* out);
typedef size_t (*huf_fn)(void* dst, size_t dst_size, const void* src, size_t src_size);
typedef int64_t (*dec_fn)(int codec, int elem_bytes, const void* src, void* dest);

uint64_t oracle_huf0_chain_chunks(void* huf, void* dec, int codec, int elem_bytes, const uint8_t* blocks, const uint64_t* block_offsets,
                                  const uint32_t* stream_sizes, uint64_t nchunks, uint32_t chunk_len, uint8_t* scratch, void* out);

typedef struct {
    /* what to run */
    int kind;                    /* 0: chunks_fn over [lo, hi); 1: Huff0 chain over [lo, hi) */
    void *f0, *f1;               /* kind 0: f0 = chunks_fn;  kind 1: f0 = huf_fn, f1 = dec_fn */
    int codec, elem_bytes;
    const uint8_t* comp;
    const uint64_t* offsets;
    const uint32_t* stream_sizes;
    uint32_t chunk_len;
    uint8_t* out;
    uint8_t* scratch;
    uint64_t lo, hi;
    int reps, cpu;
    volatile int* go;
    volatile int* ready;
    double t0, t1;
    uint64_t elems;
} job_t;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* worker(void* p)
{
    job_t* j = (job_t*)p;
    if (j->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(j->cpu, &set);
        pthread_setaffinity_np(pthread_self(), sizeof set, &set);   /* best effort */
    }
    /* this thread's slice of the compressed input, copied by the thread itself: first touch puts the pages on its own
     * NUMA node (the caller's buffer was filled by one thread and sits on one node) */
    const uint64_t n = j->hi - j->lo, b0 = j->offsets[j->lo], nbytes = j->offsets[j->hi] - b0;
    uint8_t* local = (uint8_t*)malloc((size_t)nbytes + 256);
    uint64_t* loffs = (uint64_t*)malloc((size_t)(n + 1) * sizeof(uint64_t));
    const uint8_t* comp = j->comp;
    const uint64_t* offs = j->offsets + j->lo;
    if (local && loffs) {
        for (uint64_t i = 0; i < nbytes + 64; i++) local[i] = i < nbytes ? j->comp[b0 + i] : 0;
        for (uint64_t i = 0; i <= n; i++) loffs[i] = j->offsets[j->lo + i] - b0;
        comp = local;
        offs = loffs;
    }
    __atomic_fetch_add(j->ready, 1, __ATOMIC_ACQ_REL);
    while (!__atomic_load_n(j->go, __ATOMIC_ACQUIRE)) sched_yield();
    j->t0 = now_s();
    uint64_t e = 0;
    for (int r = 0; r < j->reps; r++) {
        if (j->kind == 0)
            e = ((chunks_fn)j->f0)(j->codec, j->elem_bytes, comp, offs, n, j->chunk_len, j->out);
        else
            e = oracle_huf0_chain_chunks(j->f0, j->f1, j->codec, j->elem_bytes, comp, offs, j->stream_sizes + j->lo, n, j->chunk_len,
                                         j->scratch, j->out);
    }
    j->t1 = now_s();
    j->elems = e;
    free(local);
    free(loffs);
    return NULL;
}

/* -> seconds per pass (sustained, all threads), or a negative value on failure; *elems = elements one pass decodes.
 * out: nchunks*chunk_len*elem_bytes + nthreads*out_gap bytes; scratch (kind 1): nthreads * 65536 bytes;
 * cpus: nthreads logical CPU ids to pin to, or NULL */
static double run_mt(job_t proto, uint64_t nchunks, uint32_t out_gap, int nthreads, int reps, const int* cpus, uint64_t* elems)
{
    if (nthreads < 1 || reps < 1) return -1.0;
    if ((uint64_t)nthreads > nchunks) nthreads = (int)nchunks;
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof *th);
    job_t* jobs = (job_t*)calloc((size_t)nthreads, sizeof *jobs);
    volatile int go = 0, ready = 0;
    if (!th || !jobs) { free(th); free(jobs); return -1.0; }
    const uint64_t chunk_bytes = (uint64_t)proto.chunk_len * (uint64_t)proto.elem_bytes;
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = proto;
        jobs[t].lo = nchunks * (uint64_t)t / (uint64_t)nthreads;
        jobs[t].hi = nchunks * (uint64_t)(t + 1) / (uint64_t)nthreads;
        jobs[t].out = proto.out + jobs[t].lo * chunk_bytes + (uint64_t)t * out_gap;
        jobs[t].scratch = proto.scratch ? proto.scratch + (size_t)t * 65536 : NULL;
        jobs[t].reps = reps;
        jobs[t].cpu = cpus ? cpus[t] : -1;
        jobs[t].go = &go;
        jobs[t].ready = &ready;
        if (pthread_create(&th[t], NULL, worker, &jobs[t]) != 0) break;
        started++;
    }
    while (__atomic_load_n(&ready, __ATOMIC_ACQUIRE) < started) sched_yield();      /* every slice copied before the clock starts */
    __atomic_store_n(&go, 1, __ATOMIC_RELEASE);
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    double t0 = 0, t1 = 0;
    uint64_t e = 0;
    for (int t = 0; t < started; t++) {
        if (t == 0 || jobs[t].t0 < t0) t0 = jobs[t].t0;
        if (t == 0 || jobs[t].t1 > t1) t1 = jobs[t].t1;
        e += jobs[t].elems;
    }
    free(th);
    free(jobs);
    if (started != nthreads) return -2.0;
    if (elems) *elems = e;
    return (t1 - t0) / (double)reps;
}

double oracle_mt_decompress_chunks(void* fn, int codec, int elem_bytes, const uint8_t* comp, const uint64_t* offsets, uint64_t nchunks,
                                   uint32_t chunk_len, uint8_t* out, uint32_t out_gap, int nthreads, int reps, const int* cpus,
                                   uint64_t* elems)
{
    job_t p = {0};
    p.kind = 0; p.f0 = fn; p.codec = codec; p.elem_bytes = elem_bytes; p.comp = comp; p.offsets = offsets;
    p.chunk_len = chunk_len; p.out = out;
    return run_mt(p, nchunks, out_gap, nthreads, reps, cpus, elems);
}

double oracle_mt_huf0_chain(void* huf, void* dec, int codec, int elem_bytes, const uint8_t* blocks, const uint64_t* block_offsets,
                            const uint32_t* stream_sizes, uint64_t nchunks, uint32_t chunk_len, uint8_t* scratch, uint8_t* out,
                            uint32_t out_gap, int nthreads, int reps, const int* cpus, uint64_t* elems)
{
    job_t p = {0};
    p.kind = 1; p.f0 = huf; p.f1 = dec; p.codec = codec; p.elem_bytes = elem_bytes; p.comp = blocks; p.offsets = block_offsets;
    p.stream_sizes = stream_sizes; p.chunk_len = chunk_len; p.out = out; p.scratch = scratch;
    return run_mt(p, nchunks, out_gap, nthreads, reps, cpus, elems);
}
