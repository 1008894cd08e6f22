// Development probe (GPU box): does RCCL accept two ranks of one process on the SAME device (for a 2-rank test on a 1-GPU box)?
//   hipcc tools/rccl_probe.cpp -o tools/rccl_probe -lrccl && tools/rccl_probe
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <vector>
int main() {
    int ndev = 0; hipGetDeviceCount(&ndev);
    printf("devices: %d\n", ndev);
    int devs[2] = {0, ndev > 1 ? 1 : 0};
    ncclComm_t comms[2];
    ncclResult_t r = ncclCommInitAll(comms, 2, devs);
    printf("ncclCommInitAll({%d,%d}): %s\n", devs[0], devs[1], ncclGetErrorString(r));
    if (r != ncclSuccess) return 1;
    float *a, *b; hipStream_t s0, s1;
    hipSetDevice(devs[0]); hipMalloc(&a, 1024 * 4); hipStreamCreate(&s0);
    hipSetDevice(devs[1]); hipMalloc(&b, 1024 * 4); hipStreamCreate(&s1);
    std::vector<float> h(1024, 3.5f); hipMemcpy(a, h.data(), 4096, hipMemcpyHostToDevice);
    ncclGroupStart();
    ncclSend(a, 1024, ncclFloat, 1, comms[0], s0);
    ncclRecv(b, 1024, ncclFloat, 0, comms[1], s1);
    r = ncclGroupEnd();
    printf("group: %s\n", ncclGetErrorString(r));
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    hipMemcpy(h.data(), b, 4096, hipMemcpyDeviceToHost);
    printf("received %f\n", h[7]);
    return 0;
}
