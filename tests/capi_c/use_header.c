/* Compiled as plain C (gcc -std=c99) by tests/test_capi_cpu.py: include/svsdct.h must be a valid C header and the
 * library must link from C.  Calls only entry points that need no GPU. */
#include <stdio.h>
#include <string.h>

#include "svsdct.h"

int main(void) {
    svs_planes p;
    memset(&p, 0, sizeof p);
    p.n_frames = 600;
    p.height = 2160;
    p.width = 3840;
    p.row_pitch = 3840;
    p.frame_pitch = 3840LL * 2160;
    if (svs_abi_version() != SVS_ABI_VERSION) return 1;
    if (svs_capacity_bits(&p, 3) != 600ULL * 129600ULL * 3ULL) return 2;
    if (svs_packed_bytes(svs_capacity_bits(&p, 3)) != 29160000ULL) return 3;
    /* argument validation happens before any HIP call */
    p.width = 3836;
    {
        uint64_t n = 0;
        unsigned char dummy[8];
        int rc = svs_extract_dev(dummy, &p, 8.0, 3, dummy, 8, 0u, &n, NULL);
        if (rc != SVS_ERR_INVALID_ARG) return 4;
        if (strstr(svs_last_error(), "multiples of 8") == NULL) return 5;
    }
    printf("c abi ok, flags bit %u\n", SVS_EXACT_POCKETFFT);
    return 0;
}
