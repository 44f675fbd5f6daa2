// Instruction-throughput microbenchmark for the integer/FP64 paths a 256-bit Montgomery multiply can
// be built from on gfx950.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench tools/microbench.hip
// Prints wave-instructions per cycle per CU (4 SIMDs) at 1, 2, 4, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define KERNEL(NAME, DECL, BODY, SINK)                                              \
    __global__ void NAME(uint32_t* out, uint32_t seed) {                            \
        DECL;                                                                       \
        for (int it = 0; it < ITERS; ++it) {                                        \
            BODY BODY BODY BODY BODY BODY BODY BODY                                 \
        }                                                                           \
        SINK;                                                                       \
    }

// 8 independent chains per body instance -> 64 instr / iteration
#define DECL_U32 uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19, b = seed | 1
#define SINK_U32 if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[0] = a0
#define DECL_U64 uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; uint32_t b = seed | 1, c = seed * 7 + 3
#define SINK_U64 if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345) out[0] = (uint32_t)a0
#define DECL_F64 double a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19, b = 1.0000001, c = 0.5
#define SINK_F64 if ((a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) == 0.12345) out[0] = 1

#define OP8(ASM, CONS)  \
    asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a1) : CONS); asm volatile(ASM : "+v"(a2) : CONS); asm volatile(ASM : "+v"(a3) : CONS); \
    asm volatile(ASM : "+v"(a4) : CONS); asm volatile(ASM : "+v"(a5) : CONS); asm volatile(ASM : "+v"(a6) : CONS); asm volatile(ASM : "+v"(a7) : CONS);

#define COMMA ,
KERNEL(k_mad_u64_u32, DECL_U64, OP8("v_mad_u64_u32 %0, vcc, %1, %2, %0", "v"(b) COMMA "v"(c) : "vcc"), SINK_U64)
KERNEL(k_mul_lo_u32, DECL_U32, OP8("v_mul_lo_u32 %0, %0, %1", "v"(b)), SINK_U32)
KERNEL(k_mul_hi_u32, DECL_U32, OP8("v_mul_hi_u32 %0, %0, %1", "v"(b)), SINK_U32)
KERNEL(k_add_u32, DECL_U32, OP8("v_add_u32 %0, %0, %1", "v"(b)), SINK_U32)
KERNEL(k_add_co_u32, DECL_U32, OP8("v_add_co_u32 %0, vcc, %0, %1", "v"(b) : "vcc"), SINK_U32)
KERNEL(k_addc_co_u32, DECL_U32, OP8("v_addc_co_u32 %0, vcc, %0, %1, vcc", "v"(b) : "vcc"), SINK_U32)
KERNEL(k_add3_u32, DECL_U32, OP8("v_add3_u32 %0, %0, %1, %1", "v"(b)), SINK_U32)
KERNEL(k_mov_b32, DECL_U32, OP8("v_mov_b32 %0, %1", "v"(b)), SINK_U32)
KERNEL(k_lshl_add_u64, DECL_U64, OP8("v_lshl_add_u64 %0, %0, 0, %1", "v"((uint64_t)b)), SINK_U64)
KERNEL(k_mad_u32_u24, DECL_U32, OP8("v_mad_u32_u24 %0, %0, %1, %1", "v"(b)), SINK_U32)
KERNEL(k_mul_hi_u32_u24, DECL_U32, OP8("v_mul_hi_u32_u24 %0, %0, %1", "v"(b)), SINK_U32)
KERNEL(k_fma_f64, DECL_F64, OP8("v_fma_f64 %0, %0, %1, %2", "v"(b) COMMA "v"(c)), SINK_F64)
KERNEL(k_fma_f32, DECL_U32, OP8("v_fma_f32 %0, %0, %1, %1", "v"(b)), SINK_U32)
KERNEL(k_mad_i32_i24, DECL_U32, OP8("v_mad_i32_i24 %0, %0, %1, %1", "v"(b)), SINK_U32)
KERNEL(k_cndmask, DECL_U32, OP8("v_cndmask_b32 %0, %0, %1, vcc", "v"(b)), SINK_U32)
KERNEL(k_and_b32, DECL_U32, OP8("v_and_b32 %0, %0, %1", "v"(b)), SINK_U32)
KERNEL(k_and_imm, DECL_U32, OP8("v_and_b32 %0, 0x1fffffff, %0", "v"(b)), SINK_U32)
KERNEL(k_lshrrev_b32, DECL_U32, OP8("v_lshrrev_b32 %0, 1, %0", "v"(b)), SINK_U32)
KERNEL(k_ashrrev_i32, DECL_U32, OP8("v_ashrrev_i32 %0, 1, %0", "v"(b)), SINK_U32)
KERNEL(k_sub_u32, DECL_U32, OP8("v_sub_u32 %0, %0, %1", "v"(b)), SINK_U32)
KERNEL(k_alignbit, DECL_U32, OP8("v_alignbit_b32 %0, %0, %1, 29", "v"(b)), SINK_U32)
KERNEL(k_and_or, DECL_U32, OP8("v_and_or_b32 %0, %0, %1, %1", "v"(b)), SINK_U32)
KERNEL(k_bfe_u32, DECL_U32, OP8("v_bfe_u32 %0, %0, 1, 29", "v"(b)), SINK_U32)
KERNEL(k_lshl_add_u32, DECL_U32, OP8("v_lshl_add_u32 %0, %0, 1, %1", "v"(b)), SINK_U32)
KERNEL(k_lshrrev_b64, DECL_U64, OP8("v_lshrrev_b64 %0, 1, %0", "v"(b)), SINK_U64)
KERNEL(k_mad_sgpr, DECL_U64, OP8("v_mad_u64_u32 %0, vcc, %1, %2, %0", "v"(b) COMMA "s"(c) : "vcc"), SINK_U64)
#define DEP8(ASM, CONS)  \
    asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a0) : CONS); \
    asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a0) : CONS); asm volatile(ASM : "+v"(a0) : CONS);
KERNEL(k_mad_dep, DECL_U64, DEP8("v_mad_u64_u32 %0, vcc, %1, %2, %0", "v"(b) COMMA "v"(c) : "vcc"), SINK_U64)
KERNEL(k_mad_dep_grp, DECL_U64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0\nv_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(b) COMMA "v"(c) : "vcc");, SINK_U64)
KERNEL(k_mad_dep2, DECL_U64, asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\nv_mad_u64_u32 %1, vcc, %2, %3, %1\nv_mad_u64_u32 %0, vcc, %2, %3, %0\nv_mad_u64_u32 %1, vcc, %2, %3, %1\nv_mad_u64_u32 %0, vcc, %2, %3, %0\nv_mad_u64_u32 %1, vcc, %2, %3, %1\nv_mad_u64_u32 %0, vcc, %2, %3, %0\nv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(a0) COMMA "+v"(a1) : "v"(b) COMMA "v"(c) : "vcc");, SINK_U64)
KERNEL(k_mov_b64, DECL_U64, OP8("v_mov_b64 %0, %1", "v"((uint64_t)b)), SINK_U64)

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double mhz = prop.clockRate / 1000.0;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", prop.name, cus, mhz);
    uint32_t* d;
    CHK(hipMalloc(&d, 64));
    Entry es[] = {{"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
                  {"v_add_u32", k_add_u32}, {"v_add_co_u32", k_add_co_u32}, {"v_addc_co_u32", k_addc_co_u32},
                  {"v_add3_u32", k_add3_u32}, {"v_mov_b32", k_mov_b32}, {"v_lshl_add_u64", k_lshl_add_u64},
                  {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24}, {"v_mad_i32_i24", k_mad_i32_i24},
                  {"v_fma_f64", k_fma_f64}, {"v_fma_f32", k_fma_f32}, {"v_cndmask_b32", k_cndmask},
                  {"v_and_b32", k_and_b32}, {"v_and_b32 imm", k_and_imm}, {"v_lshrrev_b32", k_lshrrev_b32}, {"v_ashrrev_i32", k_ashrrev_i32},
                  {"v_sub_u32", k_sub_u32}, {"v_alignbit_b32", k_alignbit}, {"v_and_or_b32", k_and_or}, {"v_bfe_u32", k_bfe_u32},
                  {"v_lshl_add_u32", k_lshl_add_u32}, {"v_lshrrev_b64", k_lshrrev_b64}, {"v_mad_u64 sgpr", k_mad_sgpr}, {"v_mov_b64", k_mov_b64},
                  {"mad dep+nop", k_mad_dep}, {"mad dep grouped", k_mad_dep_grp}, {"mad 2 chains", k_mad_dep2}};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-18s %10s %10s %10s %10s   (wave-instr / clk / CU @ clockRate; cycles per wave-instr per SIMD)\n", "instr", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
    for (auto& e : es) {
        printf("%-18s", e.name);
        for (int wps = 1; wps <= 8; wps *= 2) {
            int threads = 256;                 // 4 waves = 1 per SIMD
            int blocks = cus * wps;            // wps blocks per CU
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
            CHK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
                CHK(hipEventRecord(e1, 0));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double instr_per_wave = (double)ITERS * 64.0;
            double waves_per_cu = 4.0 * wps;
            double clks = best * 1e-3 * mhz * 1e6;
            double ipc_cu = instr_per_wave * waves_per_cu / clks;
            printf(" %5.2f/%4.1f", ipc_cu, 4.0 / ipc_cu);
        }
        printf("\n");
    }
    return 0;
}
