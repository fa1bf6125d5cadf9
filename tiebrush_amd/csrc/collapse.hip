// collapse.hip — placeholder until the collapse pipeline lands
#include "tbk_internal.h"
int tbk_collapse_device(tbk_ctx* ctx, const tbk_collapse_opts*, const tbk_soa_in*, tbk_groups_out*) {
  ctx->last_error = "collapse pipeline not built";
  return TBK_EUNSUPPORTED;
}
int tbk_sample_device(tbk_ctx* ctx, const tbk_cov_in*, int32_t, tbk_sample_out*) {
  ctx->last_error = "sample pipeline not built";
  return TBK_EUNSUPPORTED;
}
extern "C" int tbk_groups_to_cov_in(tbk_ctx* ctx, const tbk_soa_in*, const tbk_groups_out*, tbk_cov_in*) {
  ctx->last_error = "not built";
  return TBK_EUNSUPPORTED;
}
