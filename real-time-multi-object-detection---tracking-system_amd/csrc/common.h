// common.h -- error plumbing and small helpers shared by every translation unit of
// librtmodt_hip.so.  gfx950 only; no CUDA compatibility layer.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rtmodt.h"

namespace rtmodt {

// thread-local last-error string behind rtmodt_last_error()
std::string &last_error();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define RT_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return ::rtmodt::fail(RTMODT_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,           \
                                  hipGetErrorString(_e));                                              \
    } while (0)

#define RT_CHECK(cond, code, ...)                                  \
    do {                                                           \
        if (!(cond)) return ::rtmodt::fail((code), __VA_ARGS__);   \
    } while (0)

#define RT_TRY(expr)               \
    do {                           \
        int _r = (expr);           \
        if (_r != RTMODT_OK) return _r; \
    } while (0)

// Run-time options.  librtmodt_hip.so reads the RTMODT_<name> variables of this ONE table and no other: each through rt_opt(), at
// detector-create / autotune time (never per launch; a running detector does not change when the environment does).
//   engine shape        CHAINS (N sub-batch chains instead of the staged engine), STAGES (1 / 2 / 3), CHAIN_JOIN (chains meet per batch),
//                       CHAIN_PROBE (0: trust stream creation order -- counter-collecting profilers serialise kernels), ZERO_COPY
//   tuner               TUNE_CACHE (file), TUNE_LOG (print every timing)
//   diagnostics         DEBUG (print the device / library banner once)
//   FORCE hooks of the parity tests (tests/test_gpu_detector.py: every kernel variant is run against the oracle, whatever the tuner
//   would pick on this box): TILE, TILE_K64, TILE_3X3S1, EPI16, TAIL, BNECK, BNECK_TAIL, UP_READ, STEM_FUSE, NO_HEAD_FINAL,
//                       NMS_THREADS, PAD_STREAMS, FRONT, BNECK32
// The A/B switches of finished experiments (profiles/r0N/README.md has their measurements) exist only in a diagnostic build
// (`make DIAG=1`, -DRTMODT_DIAG): rt_diag() is a constant nullptr otherwise and the branches behind it fold away.
static const char *const kOptions[] = {"CHAINS", "STAGES", "CHAIN_JOIN", "CHAIN_PROBE", "ZERO_COPY", "TUNE_CACHE", "TUNE_LOG", "DEBUG",
                                       "TILE", "TILE_K64", "TILE_3X3S1", "EPI16", "TAIL", "BNECK", "BNECK_TAIL", "UP_READ", "STEM_FUSE",
                                       "NO_HEAD_FINAL", "NMS_THREADS", "PAD_STREAMS", "FRONT", "BNECK32"};
static inline const char *rt_env(const char *name) {
    char buf[64];
    snprintf(buf, sizeof(buf), "RTMODT_%s", name);
    return getenv(buf);
}
// A name that is not in the table is a programming error of the LIBRARY; it must still not take the host process down: the
// lookup answers "unset", the error text goes to rtmodt_last_error(), and a sticky flag makes the next rtmodt_detector_create /
// rtmodt_option call return RTMODT_E_INVALID (engine.hip: note_bad_option / bad_option_pending).
void note_bad_option(const char *name);
static inline const char *rt_opt(const char *name) {
    for (const char *k : kOptions)
        if (strcmp(k, name) == 0) return rt_env(name);
    note_bad_option(name);
    return nullptr;
}
#if defined(RTMODT_DIAG)
static inline const char *rt_diag(const char *name) { return rt_env(name); }
#else
static inline const char *rt_diag(const char *) { return nullptr; }
#endif

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

typedef _Float16 f16;

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: raise it once per (device, size), on the device that is
// current (a process that builds detectors on two devices must not skip the second because the first already raised it).  `seen` = the
// caller's static per-kernel table.
struct DynLdsSeen { size_t bytes[64] = {}; };
static inline int raise_dynamic_lds(const void *kernel, size_t smem, DynLdsSeen &seen) {
    int dev = 0;
    RT_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && smem <= seen.bytes[dev]) return RTMODT_OK;
    RT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    if (tracked) seen.bytes[dev] = smem;
    return RTMODT_OK;
}

// SiLU of every conv epilogue: x / (1 + e^-x) as v_mul, v_exp_f32, v_add, v_rcp_f32, v_mul (the two transcendentals issue at a
// quarter of the VALU rate).  -DRTMODT_ABLATE_SILU2 (DIAGNOSTIC build) evaluates it TWICE, on x and on x + 1e-30, and returns a value within
// an ulp of the right one: same data, same decisions, twice the activation work -- which prices the activation in time and in clock
// (power) without changing what the MFMAs chew on: profiles/r03/README.md, ablations.
#if defined(__HIPCC__)
__device__ __forceinline__ float silu(float x) {
    const float a = x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#if defined(RTMODT_ABLATE_SILU2)
    const float y = x + 1e-30f;
    const float b = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
    return a + (b - a);
#else
    return a;
#endif
}
// Four SiLUs (one MFMA accumulator quad) with the plain multiplies and the add of each PAIR issued as one packed instruction:
// v_pk_mul_f32 (x * -log2 e), 2 x v_exp_f32, v_pk_add_f32 (1 + t), 2 x v_rcp_f32, v_pk_mul_f32 -- 22 issue cycles per value against
// silu()'s 26 (transcendentals 8 each, everything else 4 per instruction, packed or not).  Same operations, same roundings: the
// values are bit-identical to silu()'s (hipcc packs only the last multiply by itself: a 32-bit literal cannot be a packed operand).
typedef float floatx2_t __attribute__((ext_vector_type(2)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx2_t silu2(floatx2_t x) {
#if defined(RTMODT_ABLATE_SILU2) || defined(RTMODT_SILU_SCALAR)      // (SILU_SCALAR: check build -- tools/ab/r04_silu_bits.sh compares the stored bits of the two forms)
    return floatx2_t{silu(x[0]), silu(x[1])};
#else
    // (vector expressions, not inline asm: the hazard recognizer does not see inside an asm statement, and a VALU instruction that reads a
    //  transcendental's result one issue later reads the old register -- 1.2 % wrong outputs in the first form of this function)
    const floatx2_t k = {-1.4426950408889634f, -1.4426950408889634f}, one = {1.0f, 1.0f};
    floatx2_t t = x * k;
    t[0] = __builtin_amdgcn_exp2f(t[0]); t[1] = __builtin_amdgcn_exp2f(t[1]);
    floatx2_t d = t + one;
    d[0] = __builtin_amdgcn_rcpf(d[0]); d[1] = __builtin_amdgcn_rcpf(d[1]);
    const floatx2_t y = x * d;
    return y;
#endif
}
__device__ __forceinline__ void silu4(floatx4_t &v) {
    const floatx2_t a = silu2(floatx2_t{v[0], v[1]}), b = silu2(floatx2_t{v[2], v[3]});
    v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}
#endif

}  // namespace rtmodt
