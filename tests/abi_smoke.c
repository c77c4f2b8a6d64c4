/* Torch-free use of the C ABI: links against libhpc_rll_hip.so only (plus the HIP runtime it depends on) and calls
 * entry points that need no GPU.  Built and run by tests/test_abi.py::test_c_program_links_and_runs. */
#include <stdio.h>
#include <string.h>

#include "hpc_rll_hip.h"

int main(void) {
    if (hpc_rll_abi_version() != 6) return 1;
    if (strcmp(hpc_rll_status_string(HPC_RLL_OK), hpc_rll_status_string(HPC_RLL_EINVAL)) == 0) return 2;
    /* argument validation happens before any device work: invalid sizes come back as status codes */
    if (hpc_rll_gae_forward(NULL, NULL, NULL, NULL, -1, 4, 0.99f, NULL) != HPC_RLL_EINVAL) return 3;
    if (hpc_rll_lstm_workspace_floats(4, 2, 8, 16, 1, 0.f) <= 0) return 4;
    if (hpc_rll_partials_floats(10) < 10) return 5;
    printf("abi %d ok\n", hpc_rll_abi_version());
    return 0;
}
