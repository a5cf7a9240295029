// The library's explicit context object (tdm_hip.h "Contexts"; SURVEY.md section 8b, threading: "no global mutable state except
// an explicit tdm_ctx").  Every selector the entry points consult — arithmetic of the convolutions / GEMMs / attention, two launch
// queues in the backward, the two-part gradient reduction — and the side queue with its events are fields of a TdmCtx
// (tdm_common.h).  Entry points take the calling thread's CURRENT context: the one bound by tdm_ctx_make_current, else a default
// the thread owns.  The binding is the library's only thread-local state; a context serves one thread at a time.
#include "tdm_common.h"
#include "../../include/tdm_hip.h"

namespace {
thread_local TdmCtx t_default;          // (its side lane, if one was ever created, goes with the thread)
thread_local tdm_ctx* t_bound = nullptr;
}  // namespace

TdmCtx& tdm_cur_ctx() { return t_bound != nullptr ? t_bound->c : t_default; }

// tdm_ctx_destroy (comm.hip) calls this first.  Every backward joins its side queue into the caller's stream before it returns, so
// a context is idle between calls; deleting it destroys the side stream and events it created (~TdmSideLane).
int tdm_ctx_unbind_for_destroy(tdm_ctx* ctx) {
    if (ctx == t_bound) { t_bound = nullptr; __atomic_store_n(&ctx->c.bound, 0, __ATOMIC_RELEASE); }
    TDM_REQUIRE(__atomic_load_n(&ctx->c.bound, __ATOMIC_ACQUIRE) == 0, "ctx_destroy: the context is current on another thread");
    return 0;
}

extern "C" {

// Binds `ctx` to the calling thread (NULL: back to the thread's default context).  A context that another thread has current is
// refused: its side queue and selectors are not synchronised.
int tdm_ctx_make_current(tdm_ctx* ctx) {
    if (ctx == t_bound) return 0;
    if (ctx != nullptr) {
        int expect = 0;
        TDM_REQUIRE(__atomic_compare_exchange_n(&ctx->c.bound, &expect, 1, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE),
                    "ctx_make_current: the context is current on another thread");
    }
    if (t_bound != nullptr) __atomic_store_n(&t_bound->c.bound, 0, __ATOMIC_RELEASE);
    t_bound = ctx;
    return 0;
}

tdm_ctx* tdm_ctx_current(void) { return t_bound; }

int tdm_ctx_set_arithmetic(tdm_ctx* ctx, int conv_mode, int gemm_mode, int attn_mode) {
    TDM_REQUIRE(ctx != nullptr, "ctx_set_arithmetic: NULL context");
    TDM_REQUIRE(conv_mode == 0 || conv_mode == 2, "ctx_set_arithmetic: conv mode %d (0 exact fp32, 2 bf16x3 over S16 tensors)", conv_mode);
    TDM_REQUIRE(gemm_mode >= 0 && gemm_mode <= 2, "ctx_set_arithmetic: gemm mode %d (0 fp32, 1 bf16x3, 2 plain bf16)", gemm_mode);
    TDM_REQUIRE(attn_mode >= 0 && attn_mode <= 2, "ctx_set_arithmetic: attention mode %d (0 scalar, 1 fp32 MFMA, 2 bf16x3 MFMA)", attn_mode);
    ctx->c.conv_mode = conv_mode; ctx->c.gemm_mode = gemm_mode; ctx->c.attn_mode = attn_mode;
    return 0;
}

int tdm_ctx_get_arithmetic(const tdm_ctx* ctx, int* conv_mode, int* gemm_mode, int* attn_mode) {
    TDM_REQUIRE(ctx != nullptr, "ctx_get_arithmetic: NULL context");
    if (conv_mode) *conv_mode = ctx->c.conv_mode;
    if (gemm_mode) *gemm_mode = ctx->c.gemm_mode;
    if (attn_mode) *attn_mode = ctx->c.attn_mode;
    return 0;
}

int tdm_ctx_set_overlap(tdm_ctx* ctx, int bwd_overlap, int early_grads) {
    TDM_REQUIRE(ctx != nullptr && (bwd_overlap == 0 || bwd_overlap == 1) && (early_grads == 0 || early_grads == 1),
                "ctx_set_overlap: NULL context or a switch that is not 0 / 1");
    ctx->c.bwd_overlap = bwd_overlap; ctx->c.early_grads = early_grads;
    return 0;
}

}  // extern "C"
