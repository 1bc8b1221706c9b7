// npp_light_layout.h -- row layout of the feature-major stashes of the fused NPP_Net_light chains (csrc/npp_light.hip), shared with
// the grouped weight-gradient launch over them (csrc/npp_linear.hip: npp_light_wgrad).
#pragma once

namespace npp {

constexpr int kLW = 256;                        // hidden width
constexpr int kLPosOut = 128;                   // pos_linears.0 outputs
constexpr int kLPer = 20, kLPos = 42;           // periodic / positional input widths
constexpr int kLHp = 304;                       // [f1 (256) | x_pos (42) | 0-pad (6)]: 38 k-step groups

// feature rows of the forward stash of one candidate, in order: pre-activations z_0 .. z_3, [f1 | x_pos | 0], z_p, x_per^T
enum { LS_Z0 = 0, LS_Z1 = 256, LS_Z2 = 512, LS_Z3 = 768, LS_HP = 1024, LS_ZP = LS_HP + kLHp, LS_XP = LS_ZP + kLPosOut, LS_ROWS = LS_XP + kLPer };
// ... and of the gradient stash: d z_0 .. d z_3, d f1, d z_p, d raw^T (3 rows + 1 pad)
enum { LD_Z0 = 0, LD_Z1 = 256, LD_Z2 = 512, LD_Z3 = 768, LD_F1 = 1024, LD_ZP = 1280, LD_RAW = LD_ZP + kLPosOut, LD_ROWS = LD_RAW + 4 };

// ---- 16-bit chains (csrc/npp_light16.hip): the stashes are W-format fragment arrays (npp_layout.h wfmt_unit; 64-row workgroup tiles),
// k-step offsets (16 features each) of the arrays inside the two buffers of one candidate
constexpr int kL16KsHp = 20;                    // [f1 (16 k-steps) | x_pos (42 of 64 slots)]
// forward stash: z_0 .. z_3 (fp16 pre-activations), [f1 | x_pos] (bf16), z_p (fp16, 8 k-steps), x_per (bf16, 20 of 32 slots)
enum { L16A_Z0 = 0, L16A_HP = 64, L16A_ZP = L16A_HP + kL16KsHp, L16A_XP = L16A_ZP + 8, L16A_TOTAL = L16A_XP + 2 };
// gradient stash (bf16): d z_0 .. d z_3, d f1, d z_p (8 k-steps), d raw (3 of 32 slots)
enum { L16D_Z0 = 0, L16D_F1 = 64, L16D_ZP = 80, L16D_RAW = 88, L16D_TOTAL = 90 };

}  // namespace npp
