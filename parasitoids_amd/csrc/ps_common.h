// Shared host-side helpers of libparasitoid_hip.so: error reporting, device
// buffers, uploaded FFT plans.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/parasitoid_hip.h"
#include "fft_plan.h"

extern thread_local std::string ps_tls_error;

inline int ps_fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  ps_tls_error = buf;
  return code;
}

#define PS_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      int code_ = (e_ == hipErrorOutOfMemory) ? PS_ERR_OOM : PS_ERR_HIP;                 \
      return ps_fail(code_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                     __FILE__, __LINE__);                                                \
    }                                                                                    \
  } while (0)

#define PS_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != PS_OK) return rc_; \
  } while (0)

// Device memory goes through a small caching allocator (ps_solver.hip): an MCMC run creates
// and destroys a solver whenever the kernel extent changes, and hipMalloc/hipFree of its
// ~40 buffers cost more than the evaluation itself.  Freed blocks are kept per device (up to
// PS_POOL_GB, default 16) and handed out again for requests of about the same size.
// ps_dev_free assumes the block is idle: callers synchronise first (ps_dev_quiesce).
hipError_t ps_dev_malloc(void** p, size_t bytes);
void ps_dev_free(void* p);
void ps_dev_quiesce();   // hipDeviceSynchronize: nothing in flight may still use a block about to be freed

// growable device buffer
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return PS_OK;
    if (p) {
      ps_dev_quiesce();   // queued work may still read the old block
      ps_dev_free(p);
    }
    p = nullptr;
    cap = 0;
    hipError_t e = ps_dev_malloc((void**)&p, n * sizeof(T));
    if (e != hipSuccess)
      return ps_fail(e == hipErrorOutOfMemory ? PS_ERR_OOM : PS_ERR_HIP,
                     "device allocation of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    cap = n;
    return PS_OK;
  }
  void release() {   // callers quiesce first (destroy paths)
    if (p) ps_dev_free(p);
    p = nullptr;
    cap = 0;
  }
};

struct DevPlan {
  HostFftPlan host;
  FftProg prog;  // with device pointers
  DevBuf<cplx> tw_all;  // lo | hi | gen, contiguous
  DevBuf<uint32_t> pos, pos_phys;
  bool generic = false;
  int upload() {
    prog = host.prog;
    PS_TRY(tw_all.ensure(host.tw_all.size()));
    PS_TRY(pos.ensure(host.pos.size()));
    PS_TRY(pos_phys.ensure(host.pos_phys.size()));
    PS_HIP(hipMemcpy(tw_all.p, host.tw_all.data(), host.tw_all.size() * sizeof(cplx), hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(pos.p, host.pos.data(), host.pos.size() * 4, hipMemcpyHostToDevice));
    PS_HIP(hipMemcpy(pos_phys.p, host.pos_phys.data(), host.pos_phys.size() * 4, hipMemcpyHostToDevice));
    prog.tw_lo = tw_all.p;
    prog.tw_hi = tw_all.p + prog.n_lo;
    prog.pos = pos.p;
    prog.pos_phys = pos_phys.p;
    generic = false;
    for (int s = 0; s < prog.ns; ++s)
      if (prog.radix[s] > 9 && prog.radix[s] != 16 && prog.radix[s] != 18) generic = true;
    return PS_OK;
  }
  void release() {
    tw_all.release();
    pos.release();
    pos_phys.release();
  }
};

int ps_use_device(int device);

// internal cross-module entry points (solver <-> model, same shared library)
int ps_chain_adopt_device_kernels(ps_solver* s, int nk, const int64_t* off, const int32_t* kshape,
                                  const int* row, const int* col, const double* val);
int ps_solver_set_state_device_coo(ps_solver* s, const int* row, const int* col, const double* val,
                                   int64_t nnz, int off);
int ps_solver_dom_len_internal(ps_solver* s);
int ps_solver_device_internal(ps_solver* s);
