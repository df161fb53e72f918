// Microbenchmark: throughput of LDS atomics on gfx950 (ds_add_f32 / ds_add_u32 / ds_add_u64 / plain RMW),
// full and quarter-active waves, random addresses.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int kLdsWords = 32768;  // 128 KiB
constexpr int kIters = 4096;

__device__ inline uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE, int ACTIVE_DIV>
__global__ __launch_bounds__(1024) void k(float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < kLdsWords; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  uint32_t s = threadIdx.x * 9781u + blockIdx.x * 6271u + 1u;
  const bool active = (threadIdx.x % ACTIVE_DIV) == 0;
  float v = 1.0f + threadIdx.x * 1e-3f;
  for (int it = 0; it < kIters; ++it) {
    uint32_t a = rng(s);
    if (!active) continue;
    if (MODE == 0) atomicAdd(&lds[a % kLdsWords], v);
    else if (MODE == 1) atomicAdd((unsigned int*)&lds[a % kLdsWords], (unsigned int)it);
    else if (MODE == 2) atomicAdd((unsigned long long*)&lds[(a % (kLdsWords / 2)) * 2], (unsigned long long)it);
    else if (MODE == 3) { float* p = &lds[a % kLdsWords]; *p = *p + v; }  // racy RMW: cost reference only
    else if (MODE == 4) { float2* p = (float2*)&lds[(a % (kLdsWords / 2)) * 2]; float2 q = *p; q.x += v; q.y += v; *p = q; }
    else if (MODE == 5) { atomicAdd(&lds[(a % (kLdsWords / 2)) * 2], v); atomicAdd(&lds[(a % (kLdsWords / 2)) * 2 + 1], v); }
    else if (MODE == 6) atomicMax((int*)&lds[a % kLdsWords], (int)it);
    else if (MODE == 7) atomicAdd((double*)&lds[(a % (kLdsWords / 2)) * 2], (double)v);
    else if (MODE == 8) atomicAdd((unsigned long long*)&lds[(a % (kLdsWords / 2)) * 2], (unsigned long long)(long long)(v * (float)it));
    else if (MODE >= 10) {
      // eight atomics per random draw (addresses a ^ k*0x9e5, cheap to derive): amortises the loop's own VALU work,
      // which is what the single-atomic modes above bottom out on (~7 cycles per iteration)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t ak = (a ^ (k * 0x9e5u)) % (kLdsWords / 2) * 2;
        if (MODE == 10) atomicAdd((double*)&lds[ak], (double)v);
        else if (MODE == 11) atomicAdd((unsigned long long*)&lds[ak], (unsigned long long)it);
        else if (MODE == 12) atomicAdd((unsigned int*)&lds[ak], (unsigned int)it);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}

template <int MODE, int DIV>
void run(const char* name) {
  float* out;
  hipMalloc(&out, 4096);
  auto kern = k<MODE, DIV>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsWords * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, 1024, kLdsWords * 4>>>(out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, 1024, kLdsWords * 4>>>(out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double wave_instr_per_cu = 16.0 * kIters * ((MODE == 5) ? 2 : (MODE >= 10 ? 8 : 1));
  double cyc = ms * 1e-3 * 2.4e9 / wave_instr_per_cu;
  printf("%-44s active 1/%d: %8.3f ms  ~%7.1f cycles per wave-instruction per CU (16 waves/CU)\n", name, DIV, ms, cyc);
  hipFree(out);
}

int main() {
  run<0, 1>("ds_add_f32"); run<0, 4>("ds_add_f32");
  run<1, 1>("ds_add_u32"); run<1, 4>("ds_add_u32");
  run<2, 1>("ds_add_u64"); run<2, 4>("ds_add_u64");
  run<6, 1>("ds_max_i32");
  run<7, 1>("ds_add_f64"); run<7, 4>("ds_add_f64");
  run<8, 1>("ds_add_u64 + f32->i64 convert"); run<8, 4>("ds_add_u64 + f32->i64 convert");
  run<3, 1>("plain ds_read+add+ds_write b32 (racy)"); run<3, 4>("plain ds_read+add+ds_write b32 (racy)");
  run<4, 1>("plain ds_read+add+ds_write b64 (racy)");
  run<5, 1>("2x ds_add_f32 (pair)"); run<5, 4>("2x ds_add_f32 (pair)");
  run<10, 1>("8x ds_add_f64 per draw"); run<10, 4>("8x ds_add_f64 per draw"); run<10, 8>("8x ds_add_f64 per draw");
  run<11, 1>("8x ds_add_u64 per draw"); run<11, 4>("8x ds_add_u64 per draw"); run<11, 8>("8x ds_add_u64 per draw");
  run<12, 1>("8x ds_add_u32 per draw"); run<12, 4>("8x ds_add_u32 per draw");
  return 0;
}
