// Word + char embedding front end of the text encoder (see embed.hip).
#pragma once
#include "common.h"
#include "embed_args.h"
#include "gemm.h"

namespace hual {

// offsets (floats) of the scratch pieces inside one buffer of `total` floats
struct EmbedLayout { int CP; size_t cemb, ball, yall, dxall, dfall, total; };
EmbedLayout embed_layout(int nrows, int C, int char_dim);
int embed_cpad(int char_dim);                               // char_dim padded to a multiple of 16 (row length of cemb)
int embed_gather_tasks(const EmbedArgs& a, int nrows);      // tasks of the gather that rides in the prologue launch (gemm.h PackExtra)
int launch_embed_fwd(const EmbedArgs& a, int nrows, const DropCfg& drop, hipStream_t s);
// backward: launches everything except the filter / bias gradient product, which is returned as a job for the step's
// weight-gradient launch (embed_dw_job fills the same job without launching anything: workspace planning);
// launch_embed_unpack must run after that launch.
void embed_dw_job(const EmbedArgs& a, int nrows, DwJob* dwjob);
int launch_embed_bwd(const EmbedArgs& a, const EmbedGrads& g, int nrows, const DropCfg& drop, hipStream_t s, DwJob* dwjob, bool finish = true);
int embed_finish_blocks(const EmbedArgs& a, int nrows);      // workgroups / dynamic LDS bytes of the last step when it rides elsewhere
int embed_finish_lds(const EmbedArgs& a);
int launch_embed_unpack(const EmbedArgs& a, const EmbedGrads& g, hipStream_t s);
int embed_unpack_tasks(const EmbedArgs& a);      // tasks / char_dim padding of the unpack (for the launch it rides in: rowops.h)
int embed_unpack_cpad(const EmbedArgs& a);

}  // namespace hual
