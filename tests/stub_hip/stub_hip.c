/* Stub of the 17 HIP runtime entry points libvoxelnet_hip.so binds (nm -D --undefined-only): every call succeeds and
 * does nothing.  Preloaded (LD_PRELOAD) in front of libamdhip64 by tests/test_host_scaling.py, it turns the native step
 * executor into pure HOST work — plan, launch geometry, argument blocks, event bookkeeping — so that its cost and its
 * scaling over processes can be measured on a machine without a GPU.  Test infrastructure only: never loaded by the
 * product.  hipLaunchKernel counts the launches (vn_stub_launches). */
#include <stddef.h>
#include <stdint.h>

typedef int hipError_t;
typedef struct { unsigned x, y, z; } dim3_;
static __thread unsigned long long launches;
static char fatbin_handle[8];

unsigned long long vn_stub_launches(void) { return launches; }
hipError_t __hipPushCallConfiguration(dim3_ grid, dim3_ block, size_t shmem, void *stream) { (void)grid; (void)block; (void)shmem; (void)stream; return 0; }
hipError_t __hipPopCallConfiguration(dim3_ *grid, dim3_ *block, size_t *shmem, void **stream) {
    grid->x = grid->y = grid->z = 1; block->x = block->y = block->z = 1; *shmem = 0; *stream = 0; return 0;
}
void **__hipRegisterFatBinary(const void *data) { (void)data; return (void **)fatbin_handle; }
void __hipRegisterFunction(void **modules, const void *host_fn, char *dev_fn, const char *dev_name, unsigned threads, void *tid,
                           void *bid, void *bdim, void *gdim, int *wsize) {
    (void)modules; (void)host_fn; (void)dev_fn; (void)dev_name; (void)threads; (void)tid; (void)bid; (void)bdim; (void)gdim; (void)wsize;
}
void __hipUnregisterFatBinary(void **modules) { (void)modules; }
hipError_t hipEventCreate(void **e) { *e = (void *)1; return 0; }
hipError_t hipEventCreateWithFlags(void **e, unsigned flags) { (void)flags; *e = (void *)1; return 0; }
hipError_t hipEventDestroy(void *e) { (void)e; return 0; }
hipError_t hipEventElapsedTime(float *ms, void *a, void *b) { (void)a; (void)b; *ms = 0.f; return 0; }
hipError_t hipEventRecord(void *e, void *s) { (void)e; (void)s; return 0; }
hipError_t hipEventSynchronize(void *e) { (void)e; return 0; }
hipError_t hipFuncSetAttribute(const void *f, int attr, int value) { (void)f; (void)attr; (void)value; return 0; }
hipError_t hipGetLastError(void) { return 0; }
hipError_t hipLaunchKernel(const void *f, dim3_ grid, dim3_ block, void **args, size_t shmem, void *stream) {
    (void)f; (void)grid; (void)block; (void)args; (void)shmem; (void)stream; ++launches; return 0;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, int kind) { (void)dst; (void)src; (void)n; (void)kind; return 0; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, void *stream) { (void)dst; (void)v; (void)n; (void)stream; return 0; }
hipError_t hipStreamWaitEvent(void *s, void *e, unsigned flags) { (void)s; (void)e; (void)flags; return 0; }
