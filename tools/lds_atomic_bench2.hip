// Microbenchmark 2 (round 2): is the ~190-cycle ds_add_f32 a property of the instruction or of the FP mode?
//   variants: ds_add_f32 as compiled / with fp32 denormals flushed (MODE[5:4] = 0) / ds_add_rtn_f32 / ds_pk_add_bf16 /
//   ds_pk_add_f16, next to ds_add_u32 / ds_add_u64 / ds_add_f64, 8 atomics per random draw, full / quarter / eighth waves.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench2.hip -o gpurun_abl/lds_atomic_bench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kLdsWords = 32768;  // 128 KiB
constexpr int kIters = 2048;

__device__ inline uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE, int ACTIVE_DIV, bool FLUSH_DENORM>
__global__ __launch_bounds__(1024) void k(float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < kLdsWords; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  if (FLUSH_DENORM) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 4), 0");  // fp32 and fp64/fp16 denormals: flush
  uint32_t s = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
  const bool active = (threadIdx.x % ACTIVE_DIV) == 0;
  float v = 1.0f + threadIdx.x * 1e-3f;
  float sink = 0.f;
  for (int it = 0; it < kIters; ++it) {
    uint32_t a = rng(s);
    if (!active) continue;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const uint32_t w = (a ^ (kk * 0x9e5u)) % kLdsWords;        // 4-byte slot
      const uint32_t w2 = (w >> 1) << 1;                          // 8-byte slot
      const uint32_t addr4 = w * 4, addr8 = w2 * 4;
      if (MODE == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(addr4), "v"(v) : "memory");
      else if (MODE == 1) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr4), "v"(v) : "memory"); sink += r; }
      else if (MODE == 2) asm volatile("ds_pk_add_bf16 %0, %1" ::"v"(addr4), "v"(0x3f803f80u) : "memory");
      else if (MODE == 3) asm volatile("ds_pk_add_f16 %0, %1" ::"v"(addr4), "v"(0x3c003c00u) : "memory");
      else if (MODE == 4) asm volatile("ds_add_u32 %0, %1" ::"v"(addr4), "v"(it) : "memory");
      else if (MODE == 5) { unsigned long long x = it; asm volatile("ds_add_u64 %0, %1" ::"v"(addr8), "v"(x) : "memory"); }
      else if (MODE == 6) { double x = v; asm volatile("ds_add_f64 %0, %1" ::"v"(addr8), "v"(x) : "memory"); }
      else if (MODE == 7) asm volatile("ds_max_f32 %0, %1" ::"v"(addr4), "v"(v) : "memory");
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[5] + sink;
}

template <int MODE, int DIV, bool FL>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 4096);
  auto kern = k<MODE, DIV, FL>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWords * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kern<<<256, 1024, kLdsWords * 4>>>(out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  kern<<<256, 1024, kLdsWords * 4>>>(out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  double wave_instr_per_cu = 16.0 * kIters * 8;
  double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_cu;
  printf("%-28s %s active 1/%d: %8.3f ms  ~%7.1f cycles per wave-instruction per CU\n", name, FL ? "denorm-flush" : "default-mode", DIV, ms, cyc);
  (void)hipFree(out);
}

int main() {
  run<0, 1, false>("ds_add_f32"); run<0, 4, false>("ds_add_f32"); run<0, 8, false>("ds_add_f32");
  run<0, 1, true>("ds_add_f32"); run<0, 4, true>("ds_add_f32"); run<0, 8, true>("ds_add_f32");
  run<1, 1, false>("ds_add_rtn_f32"); run<1, 4, false>("ds_add_rtn_f32");
  run<2, 1, false>("ds_pk_add_bf16"); run<2, 4, false>("ds_pk_add_bf16");
  run<3, 1, false>("ds_pk_add_f16"); run<3, 4, false>("ds_pk_add_f16");
  run<7, 1, false>("ds_max_f32"); run<7, 4, false>("ds_max_f32");
  run<4, 1, false>("ds_add_u32"); run<4, 2, false>("ds_add_u32"); run<4, 4, false>("ds_add_u32"); run<4, 8, false>("ds_add_u32");
  run<5, 1, false>("ds_add_u64"); run<5, 2, false>("ds_add_u64"); run<5, 4, false>("ds_add_u64"); run<5, 8, false>("ds_add_u64");
  run<6, 1, false>("ds_add_f64"); run<6, 2, false>("ds_add_f64"); run<6, 4, false>("ds_add_f64"); run<6, 8, false>("ds_add_f64");
  run<6, 4, true>("ds_add_f64");
  return 0;
}
