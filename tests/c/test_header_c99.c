/* The boundary is a C ABI: include/frw.h must compile as plain C99 and link against libfrw.so from C. */
#include <stdio.h>
#include <string.h>
#include "../../include/frw.h"

int main(void)
{
    frw_layout_t L;
    frw_layout_dual_t D;
    if (frw_layout(10, &L) != FRW_OK || L.num_witness != 156724 || L.num_instance != 2049 || L.num_constraints != 162870) return 1;
    if (frw_layout(9, &L) != FRW_OK || L.num_witness != 78386 || L.num_constraints != 81460) return 2;
    if (frw_layout(8, &L) != FRW_E_INVALID_ARG) return 3;
    if (frw_layout_dual(10, &D) != FRW_OK || D.num_witness != 186 * 1024 + 4 + 52) return 4;
    if (frw_gadget_block_len(FRW_G_MOD_Q) != 29 || frw_gadget_block_len(FRW_G_NORM_BOUND_1024) != 52 || frw_gadget_block_len(9) >= 0) return 5;
    if (FRW_PK_LEN(9) != 897 || FRW_PK_LEN(10) != 1793 || FRW_SIG_LEN(9) != 666 || FRW_SIG_LEN(10) != 1280) return 6;
    if (!strstr(frw_strerror(FRW_E_NO_DEVICE), "no CPU path")) return 7;
    {
        unsigned short sig[512], pk[512], hm[512];
        if (frw_synth_triples(9, 1, 1u, 0u, sig, pk, hm) != FRW_OK || sig[0] >= 12289) return 8;
    }
    if (frw_device_count() == 0) {
        frw_ctx *ctx = (frw_ctx *)0;
        if (frw_ctx_create(0, &ctx) != FRW_E_NO_DEVICE || ctx) return 9;
        if (frw_witness_ntt_verify(ctx, 10, 1, 0, 0, 0, FRW_ENC_MONTGOMERY, 0, 0, 0, 1) != FRW_E_INVALID_ARG) return 10;
    }
    printf("frw.h: C99 ok, %d device(s)\n", frw_device_count());
    return 0;
}
