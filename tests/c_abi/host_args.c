/* The HOST half of libtdm_hip.so under AddressSanitizer + UBSan, without a GPU (sanitizers run on the CPU
 * build only).  tests/test_host_asan.py compiles csrc/*.hip host-only (`--cuda-host-only -fsanitize=address,undefined`), links
 * this driver against the result and runs it.  What executes here is everything that happens BEFORE a launch: argument
 * validation, size / layout queries, the host evaluation of the dropout hash and of Philox, the error-text plumbing.
 *   - a call that must be refused returns non-zero and leaves a message in tdm_last_error();
 *   - documented no-ops (zero-sized work) and pure queries return what the header says;
 *   - host output buffers are malloc'ed at EXACTLY the documented size, so an overrun is an ASan report.
 * A launch that gets past validation fails on this box (no device, no device code in a host-only build) and is reported as a
 * non-zero return as well: both are "refused" for the purpose of this test; what may not happen is a crash or a report. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tdm_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                                       \
    do {                                                                                   \
        if (!(cond)) { printf("FAIL %s:%d: %s   [last error: %s]\n", __FILE__, __LINE__, #cond, tdm_last_error()); ++failures; } \
    } while (0)
/* refused: non-zero return and a message */
#define REFUSED(call)                                                                      \
    do {                                                                                   \
        const long long r_ = (long long)(call);                                            \
        const char* m_ = tdm_last_error();                                                 \
        if (r_ == 0 || m_ == NULL || m_[0] == 0) { printf("FAIL %s:%d: %s returned %lld, message \"%s\"\n", __FILE__, __LINE__, #call, r_, m_ ? m_ : "(null)"); ++failures; } \
    } while (0)

int main(void) {
    printf("tdm_version %d\n", tdm_version());
    EXPECT(tdm_last_error() != NULL);

    /* ---- layout / size queries (pure host arithmetic) ---- */
    {
        int32_t* offs = malloc((TDM_UNET_NTENSOR + 1) * sizeof(int32_t));
        EXPECT(tdm_unet_param_offsets(offs) == 0);
        EXPECT(offs[0] == 0 && offs[TDM_UNET_NTENSOR] == TDM_UNET_NPARAM);
        for (int i = 0; i < TDM_UNET_NTENSOR; ++i) EXPECT(offs[i] < offs[i + 1]);
        free(offs);
        REFUSED(tdm_unet_param_offsets(NULL));
        EXPECT(tdm_unet_workspace_floats(512, 1) > tdm_unet_workspace_floats(512, 0));
        EXPECT(tdm_unet_workspace_floats(0, 1) >= 0);
        EXPECT(tdm_unet_workspace_floats(-3, 1) <= 0);
        EXPECT(tdm_unet_workspace_floats((int64_t)1 << 50, 1) <= 0);          /* absurd batch: refused, not overflowed */
        EXPECT(tdm_unet_slab_floats() > 0);
        EXPECT(tdm_unet_launch_count() > 0);
        for (int i = -2; i < tdm_unet_launch_count() + 2; ++i) {
            const char* nm = tdm_unet_launch_name(i);
            EXPECT(nm != NULL && ((i >= 0 && i < tdm_unet_launch_count()) ? nm[0] != 0 : nm[0] == 0));
        }
    }
    {
        const int D = 256, depth = 3, ffn = 2048;
        const int64_t n = tdm_tt_param_count(D, depth, ffn);
        EXPECT(n > 0);
        int64_t* offs = malloc((12 * depth + 2 + 1) * sizeof(int64_t));
        EXPECT(tdm_tt_param_offsets(D, depth, ffn, offs) == 0);
        EXPECT(offs[0] == 0 && offs[12 * depth + 2] == n);
        free(offs);
        REFUSED(tdm_tt_param_offsets(D, depth, ffn, NULL));
        EXPECT(tdm_tt_param_count(0, depth, ffn) <= 0 && tdm_tt_param_count(D, -1, ffn) <= 0 && tdm_tt_param_count(D, 100, ffn) <= 0);
        EXPECT(tdm_tt_slab_floats(D, 100, ffn) <= 0 && tdm_tt_workspace_floats(4, 16, D, 4, 100, ffn, 1) <= 0);
        REFUSED(tdm_tt_param_offsets(D, 100, ffn, (int64_t*)&n));
        EXPECT(tdm_tt_workspace_floats(256, 128, D, 4, depth, ffn, 1) > tdm_tt_workspace_floats(256, 128, D, 4, depth, ffn, 0));
        EXPECT(tdm_tt_workspace_floats(256, 128, D, 3, depth, ffn, 1) <= 0);   /* D % H != 0 */
        EXPECT(tdm_tt_slab_floats(D, depth, ffn) > 0);
        EXPECT(tdm_tt_workspace_floats((int64_t)1 << 50, 1 << 30, D, 4, depth, ffn, 1) <= 0);
        EXPECT(tdm_round_workspace_floats((int64_t)1 << 60, 50257, 256) <= 0 && tdm_round_workspace_fused_floats((int64_t)1 << 60, 50257, 256, 3) <= 0);
        EXPECT(tdm_ffn_chain_mask_count(-5, 2048) <= 0 && tdm_ffn_chain_mask_count(128, 0) <= 0);
        EXPECT(tdm_resblock_scratch_floats(-1, 28, 1, 32) <= 0 && tdm_layernorm_scratch_floats(-4) <= 0);
        EXPECT(tdm_layernorm_scratch_floats(256) > 0);
        EXPECT(tdm_resblock_scratch_floats(4, 28, 1, 32) > 0);
        EXPECT(tdm_round_workspace_floats(32768, 50257, 256) > 0);
        EXPECT(tdm_round_workspace_chunked_floats(32768, 50257, 256, 8192) > 0);
        EXPECT(tdm_round_fused_ok(32768, 50257, 256) == 1 && tdm_round_fused_ok(32768, 50257, 128) == 0);
        EXPECT(tdm_round_workspace_fused_floats(32768, 50257, 256, 3) > 0);
        EXPECT(tdm_round_workspace_fused_floats(48, 64, 256, 6) > 0);          /* more segments than 32-token blocks: clamped */
        EXPECT(tdm_ffn_chain_mask_count(32768, 2048) == (32768 / 16) * (2048 / 128) * 64);
        EXPECT(tdm_ffn_chain_mask_count(17, 32) == 2 * 1 * 64);
        EXPECT(tdm_comm_unique_id_bytes() == 128);
    }

    /* ---- host evaluations: exact-size buffers ---- */
    {
        const int64_t n = 1000;
        uint8_t* keep = malloc(n);
        EXPECT(tdm_dropout_keep_u8(0.1f, 0x1234567ull, 3, 0, n, keep) == 0);
        int kept = 0;
        for (int64_t i = 0; i < n; ++i) { EXPECT(keep[i] <= 1); kept += keep[i]; }
        EXPECT(kept > 800 && kept < 980);                                      /* P(keep) = 0.9 */
        uint8_t* keep2 = malloc(n);
        EXPECT(tdm_dropout_keep_u8(0.1f, 0x1234567ull, 3, 500, n - 500, keep2) == 0);
        EXPECT(memcmp(keep + 500, keep2, n - 500) == 0);                        /* a pure function of the flat index */
        EXPECT(tdm_dropout_keep_salted_u8(0.1f, 0x1234567ull, 7u, 3, 0, n, keep2) == 0);
        EXPECT(memcmp(keep, keep2, n) != 0);
        EXPECT(tdm_dropout_keep_u8(0.0f, 1, 0, 0, n, keep) == 0);
        for (int64_t i = 0; i < n; ++i) EXPECT(keep[i] == 1);
        EXPECT(tdm_dropout_keep_u8(0.1f, 1, 0, 0, 0, keep) == 0);              /* n = 0: nothing written */
        REFUSED(tdm_dropout_keep_u8(0.1f, 1, 0, 0, n, NULL));
        REFUSED(tdm_dropout_keep_u8(1.0f, 1, 0, 0, n, keep));
        REFUSED(tdm_dropout_keep_u8(-0.5f, 1, 0, 0, n, keep));
        REFUSED(tdm_dropout_keep_u8(0.1f, 1, 0, 0, -1, keep));
        REFUSED(tdm_dropout_keep_salted_u8(0.1f, 1, 0u, 0, 0, n, NULL));
        free(keep); free(keep2);
        uint32_t* w4 = malloc(4 * sizeof(uint32_t));
        uint32_t a[4];
        EXPECT(tdm_philox_u32_host(42, 0, 0, 7, w4) == 0);
        memcpy(a, w4, sizeof a);
        EXPECT(tdm_philox_u32_host(42, 0, 0, 7, w4) == 0 && memcmp(a, w4, sizeof a) == 0);
        EXPECT(tdm_philox_u32_host(42, 1, 0, 7, w4) == 0 && memcmp(a, w4, sizeof a) != 0);
        REFUSED(tdm_philox_u32_host(42, 0, 0, 7, NULL));
        free(w4);
    }

    /* ---- arithmetic selectors ---- */
    {
        const int c = tdm_get_conv_mode(), g = tdm_get_gemm_mode(), at = tdm_get_attn_mode();
        REFUSED(tdm_set_conv_mode(1));     /* the staging-split kernels are no longer built */
        REFUSED(tdm_set_conv_mode(7));
        REFUSED(tdm_set_gemm_mode(-1));
        REFUSED(tdm_set_gemm_mode(3));
        REFUSED(tdm_set_attn_mode(9));
        EXPECT(tdm_get_conv_mode() == c && tdm_get_gemm_mode() == g && tdm_get_attn_mode() == at);
        EXPECT(tdm_set_conv_mode(0) == 0 && tdm_get_conv_mode() == 0 && tdm_set_conv_mode(c) == 0);
        EXPECT(tdm_attn_set_ablate(0) == 0 && tdm_ffn_chain_set_ablate(0) == 0);
    }

    /* ---- entry points that launch: every one refuses NULL operands / impossible shapes before touching the device ---- */
    {
        float dummy[64];          /* never dereferenced on the host: stands for "some device pointer" */
        int64_t idummy[8];
        float* p = dummy; int64_t* ip = idummy;
        REFUSED(tdm_q_sample_f32(NULL, NULL, NULL, NULL, NULL, NULL, 4, 784, NULL));
        REFUSED(tdm_q_sample_f32(p, p, ip, p, p, p, -1, 784, NULL));
        REFUSED(tdm_to_unit_u8_f32(NULL, NULL, NULL, 16, NULL));
        REFUSED(tdm_unet_fwd_f32(NULL, NULL, NULL, NULL, NULL, 4, 0, NULL));
        REFUSED(tdm_unet_fwd_f32(p, p, ip, p, p, -4, 0, NULL));
        REFUSED(tdm_unet_bwd_f32(NULL, NULL, NULL, NULL, NULL, NULL, 4, NULL));
        REFUSED(tdm_unet_loss_grad_f32(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 4, NULL));
        REFUSED(tdm_unet_loss_grad_philox_f32(NULL, NULL, NULL, NULL, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 4, NULL));
        REFUSED(tdm_unet_loss_grad_philox_epoch_f32(NULL, NULL, NULL, NULL, NULL, 64, 16, 0, NULL, NULL, 1, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 16, NULL));
        REFUSED(tdm_unet_loss_grad_philox_epoch_f32(p, p, ip, ip, ip, 64, 8, 0, p, p, 1, ip, ip, p, p, p, p, p, p, p, p, 16, NULL));   /* stride < B */
        REFUSED(tdm_unet_p_sample_step_philox_f32(NULL, NULL, NULL, NULL, NULL, NULL, 1, NULL, NULL, NULL, NULL, 4, NULL));
        REFUSED(tdm_unet_get_activation(NULL, 4, 0, NULL, NULL));
        REFUSED(tdm_unet_get_activation(p, 4, 99, p, NULL));
        REFUSED(tdm_unet_relu_mask_io(NULL, 4, 0, 1, NULL, 0, NULL));
        REFUSED(tdm_unet_relu_mask_io(p, 4, 7, 1, (uint8_t*)p, 0, NULL));
        REFUSED(tdm_unet_replay_launch_f32(p, p, ip, p, p, p, p, p, p, 4, -1, NULL));
        REFUSED(tdm_adamw_flat_f32(NULL, NULL, NULL, NULL, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1, 1.f, NULL));
        REFUSED(tdm_adamw_flat_devstep_f32(NULL, NULL, NULL, NULL, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, NULL, 1.f, NULL));
        REFUSED(tdm_adamw_flat_devsched_f32(NULL, NULL, NULL, NULL, 16, NULL, 0, 0.9f, 0.999f, 1e-8f, 0.01f, NULL, 1.f, NULL, 1, NULL));
        REFUSED(tdm_adamw_flat_devsched_f32(p, p, p, p, 16, p, 0, 0.9f, 0.999f, 1e-8f, 0.01f, ip, 1.f, NULL, 1, NULL));   /* empty lr table */
        REFUSED(tdm_mse_fwd_bwd_f32(NULL, NULL, NULL, NULL, NULL, 16, NULL));
        REFUSED(tdm_philox_normal_f32(1, 0, NULL, 16, NULL));
        REFUSED(tdm_philox_normal_f32(1, 0, p, 3, NULL));                       /* n % 4 != 0 */
        REFUSED(tdm_tt_fwd_f32(NULL, NULL, NULL, NULL, NULL, 2, 16, 32, 4, 1, 64, 0, 0.f, 0, NULL));
        REFUSED(tdm_tt_fwd_f32(p, p, ip, p, p, 2, 16, 30, 4, 1, 64, 0, 0.f, 0, NULL));   /* D % H != 0 */
        REFUSED(tdm_tt_fwd_f32(p, p, ip, p, p, 2, 16, 32, 4, 1, 64, 0, 1.5f, 0, NULL));  /* p_drop out of range */
        REFUSED(tdm_attention_fwd_f32(NULL, NULL, NULL, 2, 16, 32, 4, 0.f, 0, 0, NULL));
        REFUSED(tdm_attention_fwd_f32(p, p, p, 2, 16, 30, 4, 0.f, 0, 0, NULL));
        REFUSED(tdm_attention_bwd_f32(p, p, p, p, NULL, p, 2, 16, 32, 4, 0.f, 0, 0, NULL));
        REFUSED(tdm_attention_step_form_f32(3, p, p, p, p, p, p, p, 2, 16, 32, 4, 0.f, 0, 0, NULL));
        REFUSED(tdm_attention_step_form_f32(1, p, NULL, p, p, p, p, p, 2, 16, 32, 4, 0.f, 0, 0, NULL));   /* dQ needs O */
        REFUSED(tdm_attention_step_form_f32(0, p, NULL, NULL, NULL, NULL, p, p, 2, 16, 32, 4, 0.f, 0, 0, NULL));
        REFUSED(tdm_layernorm_residual_fwd_f32(NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 8, 32, NULL));
        REFUSED(tdm_layernorm_residual_fwd_f32(p, p, p, p, p, p, NULL, NULL, 8, 32, NULL));   /* s without mean / rstd */
        REFUSED(tdm_layernorm_residual_fwd_f32(p, p, p, p, p, NULL, NULL, NULL, 8, 30, NULL));  /* D % 4 != 0 */
        REFUSED(tdm_ffn_chain_f32(1, 3, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 1.f, 0.1f, 1, 3, 4, 128, 256, 2048, NULL));
        REFUSED(tdm_ffn_chain_f32(0, 3, p, p, p, p, p, p, NULL, NULL, 1.f, 0.f, 1, 3, 4, 128, 128, 2048, NULL));   /* D != 256 */
        REFUSED(tdm_ffn_chain_f32(0, 2, p, p, p, p, p, p, NULL, NULL, 1.f, 0.f, 1, 3, 4, 128, 256, 2048, NULL));   /* nprod */
        REFUSED(tdm_ffn_chain_f32(5, 3, p, p, p, p, p, p, NULL, NULL, 1.f, 0.f, 1, 3, 4, 128, 256, 2048, NULL));   /* mode */
        REFUSED(tdm_round_ce_loss_grad_fused_f32(NULL, NULL, NULL, NULL, 1.f, NULL, NULL, NULL, NULL, NULL, 64, 100, 256, 1, NULL));
        REFUSED(tdm_round_ce_loss_grad_fused_f32(p, p, p, ip, 1.f, p, p, p, p, p, 64, 100, 128, 1, NULL));         /* D != 256 */
        REFUSED(tdm_embed_gather_f32(NULL, NULL, NULL, 8, 100, 32, NULL));
        REFUSED(tdm_embed_scatter_add_f32(NULL, NULL, NULL, 8, 100, 32, 1.f, NULL));
        REFUSED(tdm_split_s16_f32(NULL, NULL, 64, NULL));
        REFUSED(tdm_split_s16_f32(p, p, 10, NULL));                             /* not whole 16-element groups */
        REFUSED(tdm_text_combine_dx0_f32(NULL, NULL, NULL, NULL, NULL, NULL, 8, 32, NULL));
        REFUSED(tdm_text_loss_f32(NULL, NULL, NULL, NULL, NULL, NULL));
    }

    /* ---- communicator plumbing without a device ---- */
    {
        REFUSED(tdm_ctx_create(0, NULL));
        REFUSED(tdm_comm_unique_id(NULL));
        REFUSED(tdm_comm_init(NULL, NULL, 0, 1));
        EXPECT(tdm_comm_rank(NULL) == 0 && tdm_comm_world(NULL) == 1);   /* no communicator = the single-rank defaults */
        REFUSED(tdm_allreduce_sum_f32(NULL, NULL, 16, NULL));
        REFUSED(tdm_broadcast_f32(NULL, NULL, 16, 0, NULL));
        EXPECT(tdm_ctx_destroy(NULL) == 0 || tdm_last_error()[0] != 0);
    }

    /* ---- contexts: selector state is a field of an explicit object; the set_* shorthands act on the CURRENT one ---- */
    {
        tdm_ctx *a = NULL, *b = NULL;
        int cm = -1, gm = -1, am = -1;
        EXPECT(tdm_ctx_current() == NULL);                                       /* a fresh thread works on its own default */
        EXPECT(tdm_ctx_create(0, &a) == 0 && tdm_ctx_create(0, &b) == 0);
        EXPECT(tdm_ctx_get_arithmetic(a, &cm, &gm, &am) == 0 && cm == 2 && gm == 1 && am == 2);
        REFUSED(tdm_ctx_set_arithmetic(a, 1, 1, 2));                             /* conv mode 1 is no longer built */
        REFUSED(tdm_ctx_set_arithmetic(NULL, 2, 1, 2));
        REFUSED(tdm_ctx_set_overlap(a, 2, 0));
        EXPECT(tdm_ctx_set_arithmetic(a, 0, 0, 1) == 0);
        EXPECT(tdm_get_conv_mode() == 2 && tdm_get_gemm_mode() == 1);            /* ... and the thread's default is untouched */
        EXPECT(tdm_ctx_make_current(a) == 0 && tdm_ctx_current() == a);
        EXPECT(tdm_get_conv_mode() == 0 && tdm_get_gemm_mode() == 0 && tdm_get_attn_mode() == 1);
        EXPECT(tdm_set_gemm_mode(2) == 0);                                       /* the shorthand writes the bound context */
        EXPECT(tdm_ctx_get_arithmetic(a, &cm, &gm, &am) == 0 && gm == 2);
        EXPECT(tdm_ctx_make_current(b) == 0 && tdm_get_gemm_mode() == 1);        /* b still has the defaults */
        EXPECT(tdm_ctx_make_current(NULL) == 0 && tdm_ctx_current() == NULL && tdm_get_conv_mode() == 2);
        EXPECT(tdm_ctx_make_current(a) == 0);
        EXPECT(tdm_ctx_destroy(a) == 0 && tdm_ctx_current() == NULL);            /* destroying the bound context unbinds it */
        EXPECT(tdm_ctx_destroy(b) == 0);
    }

    if (failures) { printf("%d host-side check(s) failed\n", failures); return 1; }
    printf("host args OK\n");
    return 0;
}
