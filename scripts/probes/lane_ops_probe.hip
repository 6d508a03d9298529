// Stand-alone probe (not part of the library): which lane does each lane read from under the gfx950 cross-lane forms the
// transposed butterfly (rp_common.h: rp_rows_level) can use instead of ds_bpermute?  Data = lane id.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lane_ops scripts/probes/lane_ops_probe.hip && /tmp/lane_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const unsigned A = l, B = 100 + l;
    uint2v r32 = __builtin_amdgcn_permlane32_swap(A, B, false, false);
    uint2v r16 = __builtin_amdgcn_permlane16_swap(A, B, false, false);
    out[0 * 64 + l] = r32.x; out[1 * 64 + l] = r32.y;
    out[2 * 64 + l] = r16.x; out[3 * 64 + l] = r16.y;
    out[4 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0x128, 0xf, 0xf, false);   // row_ror:8
    out[5 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    out[6 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    out[7 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0x104, 0xf, 0xf, false);   // row_shl:4
    out[8 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0x114, 0xf, 0xf, false);   // row_shr:4
    // xor 4 from two masked shifts: banks 0, 2 (lane & 4 == 0) read lane + 4, banks 1, 3 read lane - 4
    unsigned x4 = __builtin_amdgcn_update_dpp(999u, A, 0x104, 0xf, 0x5, false);
    x4 = __builtin_amdgcn_update_dpp(x4, A, 0x114, 0xf, 0xa, false);
    out[9 * 64 + l] = x4;
    unsigned y4 = __builtin_amdgcn_update_dpp(999u, A, 0x114, 0xf, 0x5, false);
    y4 = __builtin_amdgcn_update_dpp(y4, A, 0x104, 0xf, 0xa, false);
    out[10 * 64 + l] = y4;
    out[11 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0x124, 0xf, 0xf, false);  // row_ror:4
    out[12 * 64 + l] = __builtin_amdgcn_update_dpp(999u, A, 0x12c, 0xf, 0xf, false);  // row_ror:12
}

int main() {
    unsigned* d; hipMalloc(&d, 13 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[13 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"permlane32_swap.x(A)", "permlane32_swap.y(B)", "permlane16_swap.x(A)", "permlane16_swap.y(B)", "row_ror:8", "quad[1,0,3,2]",
                           "quad[2,3,0,1]", "row_shl:4", "row_shr:4", "shl4@banks0,2+shr4@banks1,3", "shr4@banks0,2+shl4@banks1,3", "row_ror:4", "row_ror:12"};
    for (int r = 0; r < 13; ++r) {
        printf("%-30s", names[r]);
        for (int l = 0; l < 64; ++l) printf(" %u", h[r * 64 + l]);
        printf("\n");
    }
    return 0;
}
