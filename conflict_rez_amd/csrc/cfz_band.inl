// Band storage of the collocation plans' KKT matrix (LAPACK gb layout: kl = ku = kb, ld = 3 kb + 1 rows per column) and the pointer types
// the out-of-line eliminations of cfz_colloc.inl and the Riccati sweeps of cfz_plan.inl take.  (Rounds 1-3 also kept a one-wavefront
// elimination in an LDS window here, for state_ws and the one-wavefront collocation kernel: both are gone, docs/notebook.md round 4.)
#pragma once

namespace cfzb {

struct Band { double *ab; int kb, ld, off; };  // entry (i, j) at ab[j * ld + off + i - j]; the band eliminations: off = 2 kb, ld = 3 kb + 1 (room for the fill); the structured ones only read it: off = kb, ld = 2 kb + 1

#if defined(__HIP_DEVICE_COMPILE__)
// The eliminations are functions of their own (registers of their own: inlined into a solver their loops reloaded spilled values from
// scratch at every pivot).  On this toolchain (gfx950, ROCm 7.2) such a function must not NAME any LDS (cfz_colloc.inl's header): the
// LDS arrays come in as address-space-3 pointers from an inlined wrapper that names them, the band as an address-space-1 pointer so
// that it is read with GLOBAL rather than FLAT instructions ...
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(1))) double glb_f64;
typedef __attribute__((address_space(1))) int glb_i32;
// ... and the pointer must reach the function as a VALUE the compiler cannot see through: when every call site of such a function passes
// the same LDS array, interprocedural constant propagation moves the array's name into the function after all, which then finds it
// through `llvm.amdgcn.(dyn)lds.offset.table` (round 4, docs/notebook.md).
template <class T>
__device__ inline T *opaque(T *p) { asm volatile("" : "+v"(p)); return p; }
#endif

}  // namespace cfzb
