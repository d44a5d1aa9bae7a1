// Probe (round 3): same-XCD producer -> consumer hand-off of a 16 KiB payload through L2, the pattern of the one-pass attention
// backward's dq chain.  Workgroups read HW_REG_XCC_ID; on every XCD the first arriver of pair p becomes the producer, the second the
// consumer (a ticket per (xcd, pair)).  Producer: every wave stores its 4 KiB part (dwordx4 per lane), s_waitcnt vmcnt(0), then lane 0
// stores the wave's flag.  Consumer wave w polls flag w, then loads part w and counts poison words.  Variants (argv[1] bit mask):
//   1 payload stores sc1      2 payload loads plain (default sc1)     4 flag store sc1 (default plain)   8 flag poll plain (default sc1)
//   16 producer: agent release fence before the flag     32 consumer: agent acquire fence after the flag
//   64 producer delays its stores by ~2 us (so that the consumer really spins)
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/handoff_l2.hip -o tools/probes/handoff_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int PAIRS = 16;  // pairs per XCD
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float* payload, unsigned* flags, unsigned* tickets, unsigned* bad, int mode, int rep) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
  __shared__ unsigned tk;
  if (tid == 0) tk = atomicAdd(tickets + xcc, 1u);
  __syncthreads();
  const unsigned t = tk;
  if (t >= 2 * PAIRS) return;
  const unsigned pair = t >> 1, role = t & 1;  // 0 producer, 1 consumer
  float* part = payload + (((size_t)xcc * PAIRS + pair) * 4 + wave) * 1024 + lane * 4;
  unsigned* flag = flags + ((size_t)xcc * PAIRS + pair) * 4 + wave;
  if (role == 0) {
    if (mode & 64) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < 200) __builtin_amdgcn_s_sleep(8); }
    const f4 v = {(float)rep, 1.f, 2.f, 3.f};
    if (mode & 1) {
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off offset:1024 sc1\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off offset:2048 sc1\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off offset:3072 sc1\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
    } else {
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off offset:1024\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off offset:2048\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
      asm volatile("global_store_dwordx4 %0, %1, off offset:3072\n\ts_nop 1" ::"v"(part), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (mode & 16) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      if (mode & 4) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(flag), "v"(1u) : "memory");
      else asm volatile("global_store_dword %0, %1, off" ::"v"(flag), "v"(1u) : "memory");
    }
  } else {
    unsigned fl = 0, spins = 0;
    for (;;) {
      if (mode & 8) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(fl) : "v"(flag) : "memory");
      else asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(fl) : "v"(flag) : "memory");
      if (__builtin_amdgcn_readfirstlane(fl) >= 1u || ++spins > (1u << 20)) break;
      __builtin_amdgcn_s_sleep(8);
    }
    if (mode & 32) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    f4 ld[4];
    if (mode & 2) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[0]) : "v"(part) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(ld[1]) : "v"(part) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(ld[2]) : "v"(part) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(ld[3]) : "v"(part) : "memory");
    } else {
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(ld[0]) : "v"(part) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:1024 sc1" : "=v"(ld[1]) : "v"(part) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:2048 sc1" : "=v"(ld[2]) : "v"(part) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:3072 sc1" : "=v"(ld[3]) : "v"(part) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3])::"memory");
    bool b = false;
    for (int i = 0; i < 4; i++) b |= ld[i][0] != (float)rep;
    if (__any(b) && lane == 0) atomicAdd(bad + (spins > (1u << 20) ? 1 : 0), 1u);
    if (lane == 0 && spins > 0) atomicAdd(bad + 2, 1u);  // consumer waves that really waited
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, reps = argc > 2 ? atoi(argv[2]) : 200;
  float* payload; unsigned *flags, *tickets, *bad;
  const size_t pf = (size_t)8 * PAIRS * 4 * 1024;
  hipMalloc(&payload, pf * 4); hipMalloc(&flags, 8 * PAIRS * 4 * 4); hipMalloc(&tickets, 64); hipMalloc(&bad, 16);
  hipMemset(bad, 0, 16);
  for (int r = 1; r <= reps; r++) {
    hipMemsetAsync(payload, 0xFF, pf * 4, 0); hipMemsetAsync(flags, 0, 8 * PAIRS * 4 * 4, 0); hipMemsetAsync(tickets, 0, 64, 0);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, payload, flags, tickets, bad, mode, r);
  }
  hipDeviceSynchronize();
  unsigned h[4];
  hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
  printf("mode %3d: %u stale consumer waves (+%u poll time-outs) of %d; %u waves really waited\n", mode, h[0], h[1], reps * 8 * PAIRS * 4, h[2]);
  return 0;
}
