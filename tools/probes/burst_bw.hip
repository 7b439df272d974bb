// probe: the decoder's WRITE PATTERN without the decoder.  131 072 chunks of 10 240 bytes; a wavefront owns 64/LANES consecutive chunks and LANES lanes
// write one chunk, 16 bytes a lane per store; 16 wavefronts a CU, every chunk written front to back in BURSTS of `burst` bytes with a pause
// between bursts that stands for the arithmetic of the rows in the burst (pause proportional to the burst: the same total pause whatever the
// burst).  decode_fast is LANES = 8, burst = 256 (16 rows of 8 uint16 columns per step).  Question: does the rate HBM takes the bytes at depend
// on how many bytes a chunk receives at once?
//   ./burst_bw  -> table
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int LANES, bool NT>
__global__ void __launch_bounds__(256) chunk_writer(uint8_t* __restrict__ dst, uint32_t nchunks, uint32_t chunk_bytes, uint32_t burst, uint32_t pause_per_256)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t chunk = wave * (64 / LANES) + lane / LANES, lane_d = lane % LANES;
    if (chunk >= nchunks) return;
    uint8_t* base = dst + (uint64_t)chunk * chunk_bytes + lane_d * 16;
    v4u o = {chunk, lane, 3, 4};
    for (uint32_t off = 0; off < chunk_bytes; off += burst) {
        for (uint32_t p = 0; p < pause_per_256 * (burst / 256); p++) __builtin_amdgcn_s_sleep(8);     // 8 x 64 clocks
        for (uint32_t j = 0; j < burst && off + j < chunk_bytes; j += LANES * 16) {
            o.x += j;
            if (NT) __builtin_nontemporal_store(o, (v4u*)(base + off + j)); else *(v4u*)(base + off + j) = o;
        }
    }
}

template <int LANES, bool NT> void run(uint8_t* dst, uint32_t burst, uint32_t pause, const char* what)
{
    const uint32_t nchunks = 131072, chunk_bytes = 10240;
    const unsigned grid = nchunks / (4 * (64 / LANES));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((chunk_writer<LANES, NT>), dim3(grid), dim3(256), 0, 0, dst, nchunks, chunk_bytes, burst, pause);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((chunk_writer<LANES, NT>), dim3(grid), dim3(256), 0, 0, dst, nchunks, chunk_bytes, burst, pause);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double b = (double)nchunks * chunk_bytes;
    printf("%-10s lanes/chunk %2d  burst %5u B  pause %2u: %.4f ms -> %.2f TB/s\n", what, LANES, burst, pause, ms, b / ms / 1e9);
}

int main()
{
    uint8_t* dst;
    const uint64_t bytes = 131072ull * 10240;
    hipMalloc(&dst, bytes + 4096); hipMemset(dst, 0, bytes);
    for (uint32_t pause : {0u, 2u, 4u}) {
        for (uint32_t burst : {256u, 512u, 1024u, 2560u, 10240u}) run<8, true>(dst, burst, pause, "nt");
        printf("\n");
    }
    for (uint32_t burst : {256u, 1024u, 10240u}) run<8, false>(dst, burst, 2, "plain");
    for (uint32_t burst : {1024u, 2048u, 10240u}) run<64, true>(dst, burst, 2, "nt");
    for (uint32_t burst : {256u, 1024u, 10240u}) run<16, true>(dst, burst, 2, "nt");
    return 0;
}
