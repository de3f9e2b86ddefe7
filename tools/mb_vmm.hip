// mb_vmm.hip — what the HIP virtual-memory API does on this box, probed before the KV cache relied on it
// (kh_model_load.hip::kv_ensure; profiles/r6_mb_vmm.txt).
//   part 1  mappings of MIXED sizes inside one reservation: which of hipMemCreate / hipMemMap / hipMemSetAccess fails
//           for which (offset, length) sequence - on ROCm 7.2 hipMemSetAccess returns "invalid argument" for many
//           mappings whose size differs from earlier ones (the first version of kv_ensure mapped one handle per run of
//           chunks and failed exactly there);
//   part 2  mappings of ONE size (8 / 2 / 32 MiB) in random order over two interleaved 16-GiB reservations, unmapped and
//           mapped again: never fails, 13-14 us per chunk - the form kv_ensure uses (and PyTorch's expandable segments).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-deprecated-declarations tools/mb_vmm.hip -o kuiperllama_amd/lib/mb_vmm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>
static int part1() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t MiB = 1 << 20;
  for (int variant = 0; variant < 2; ++variant) {
    const size_t R = 8192 * MiB;
    void* base = nullptr;
    hipError_t e = hipMemAddressReserve(&base, R, variant ? 0 : 8 * MiB, nullptr, 0);
    printf("== reservation %d: align arg %s -> %s at %p\n", variant, variant ? "0" : "8 MiB", hipGetErrorString(e), base);
    auto map = [&](size_t off_mib, size_t len_mib) {
      char* va = (char*)base + off_mib * MiB;
      hipMemGenericAllocationHandle_t h;
      hipError_t e1 = hipMemCreate(&h, len_mib * MiB, &prop, 0), e2 = hipSuccess, e3 = hipSuccess;
      if (e1 == hipSuccess) e2 = hipMemMap(va, len_mib * MiB, 0, h, 0);
      if (e1 == hipSuccess && e2 == hipSuccess) e3 = hipMemSetAccess(va, len_mib * MiB, &acc, 1);
      printf("  +%5zu MiB len %4zu MiB: create %s | map %s | set-access %s\n", off_mib, len_mib, hipGetErrorString(e1),
             e1 == hipSuccess ? hipGetErrorString(e2) : "-", (e1 == hipSuccess && e2 == hipSuccess) ? hipGetErrorString(e3) : "-");
      (void)hipGetLastError();
    };
    map(0, 8);       // at the reservation base
    map(8, 8);       // adjacent, same size
    map(16, 32);     // adjacent, larger
    map(48, 8);      // adjacent behind a larger run
    map(64, 8);      // gap
    map(100, 8);     // gap, 4 MiB-aligned only
    map(256, 32);
    map(288, 32);    // adjacent behind 32
    map(320, 8);     // adjacent behind 32
    map(1024, 64);
    map(2048, 128);
    map(3072, 240);
    map(4096, 256);
    map(5120, 512);
    map(6144, 1024);
    // set-access once over several adjacent maps?
    {
      char* va = (char*)base + 7168 * MiB;
      hipMemGenericAllocationHandle_t h1, h2;
      hipError_t a = hipMemCreate(&h1, 8 * MiB, &prop, 0), b = hipMemCreate(&h2, 8 * MiB, &prop, 0);
      hipError_t c = hipMemMap(va, 8 * MiB, 0, h1, 0), d = hipMemMap(va + 8 * MiB, 8 * MiB, 0, h2, 0);
      hipError_t f = hipMemSetAccess(va, 16 * MiB, &acc, 1);
      printf("  two maps, one set-access over both: %s %s %s %s | %s\n", hipGetErrorString(a), hipGetErrorString(b), hipGetErrorString(c),
             hipGetErrorString(d), hipGetErrorString(f));
      (void)hipGetLastError();
    }
  }
  return 0;
}

static int part2() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t MiB = 1 << 20;
  for (size_t chunk_mib : {8, 2, 32}) {
    const size_t R = 16384 * MiB, chunk = chunk_mib * MiB;
    void *b1 = nullptr, *b2 = nullptr;
    if (hipMemAddressReserve(&b1, R, 0, nullptr, 0) != hipSuccess || hipMemAddressReserve(&b2, R, 0, nullptr, 0) != hipSuccess) {
      printf("reserve failed\n");
      return 1;
    }
    const size_t nch = R / chunk;
    std::vector<size_t> order(nch);
    for (size_t i = 0; i < nch; ++i) order[i] = i;
    srand(7);
    std::random_shuffle(order.begin(), order.end());
    const size_t N = std::min<size_t>(1500, nch);
    struct Run { char* va; hipMemGenericAllocationHandle_t h; };
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<Run> runs;
      int fails[3] = {0, 0, 0};
      auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < N; ++i) {
        char* va = (char*)((i & 1) ? b2 : b1) + order[i] * chunk;  // the two reservations interleaved (K and V caches)
        hipMemGenericAllocationHandle_t h;
        hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) { ++fails[0]; (void)hipGetLastError(); continue; }
        e = hipMemMap(va, chunk, 0, h, 0);
        if (e != hipSuccess) { ++fails[1]; (void)hipGetLastError(); (void)hipMemRelease(h); continue; }
        e = hipMemSetAccess(va, chunk, &acc, 1);
        if (e != hipSuccess) {
          if (fails[2] < 3) printf("    set-access failed at i %zu, chunk index %zu\n", i, order[i]);
          ++fails[2];
          (void)hipGetLastError();
          (void)hipMemUnmap(va, chunk);
          (void)hipMemRelease(h);
          continue;
        }
        runs.push_back({va, h});
      }
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      // touch everything that was mapped
      hipError_t te = hipSuccess;
      for (auto& r : runs) if (te == hipSuccess) te = hipMemsetAsync(r.va, 1, chunk, 0);
      if (te == hipSuccess) te = hipDeviceSynchronize();
      printf("chunk %2zu MiB pass %d: %zu maps in random order over two 16-GiB reservations: create/map/set-access failures %d/%d/%d, %.1f us per chunk, memset of all: %s\n",
             chunk_mib, pass, N, fails[0], fails[1], fails[2], ms * 1e3 / N, hipGetErrorString(te));
      for (auto& r : runs) {
        (void)hipMemUnmap(r.va, chunk);
        (void)hipMemRelease(r.h);
      }
    }
    (void)hipMemAddressFree(b1, R);
    (void)hipMemAddressFree(b2, R);
  }
  return 0;
}

int main() {
  printf("## part 1: mixed mapping sizes\n");
  part1();
  printf("## part 2: one mapping size, random order\n");
  return part2();
}
