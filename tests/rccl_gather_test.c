/* GPU-box check of fiasco_amd_rccl_gather() (include/libfiasco_amd_hip.h) from plain C: a communicator of ONE rank
 * (a 1-GPU box cannot hold two), three byte strings of different lengths incl. an empty one.  Built and run by
 * tests/test_gpu_parity.py::test_rccl_gather_from_c.  Prints "ok" on success. */
#define __HIP_PLATFORM_AMD__ 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "libfiasco_amd.h"
#include "libfiasco_amd_hip.h"

int main(void)
{
    ncclUniqueId id;
    ncclComm_t comm;
    const unsigned char a[] = "FIASCO\nfirst stream", c[] = "third";
    const unsigned char *data[3] = { a, (const unsigned char *) "", c };
    const size_t len[3] = { sizeof a, 0, sizeof c };
    unsigned char **all = NULL;
    size_t *all_len = NULL;
    unsigned n_all = 0, i;
    if (hipSetDevice(0) != hipSuccess) { puts("no device"); return 2; }
    if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { puts("no communicator"); return 2; }
    if (!fiasco_amd_rccl_gather(comm, NULL, 0, 1, 0, 3, data, len, &all, &all_len, &n_all)) {
        printf("gather failed: %s\n", fiasco_get_error_message());
        return 1;
    }
    if (n_all != 3) { printf("n_all %u\n", n_all); return 1; }
    for (i = 0; i < 3; i++)
        if (all_len[i] != len[i] || memcmp(all[i], data[i], len[i]) != 0) { printf("stream %u differs\n", i); return 1; }
    for (i = 0; i < 3; i++) fiasco_amd_free(all[i]);
    free(all); free(all_len);
    ncclCommDestroy(comm);
    puts("ok");
    return 0;
}
