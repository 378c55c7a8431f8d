// The four-wave tiles of the row-gather convolution whose K loop is the assembly of kloop4w.inc (tools/gen_kloop4w.py): 256 x 256 and
// 256 x 128, bf16 / f16 / split precision; the 256 x 256 tile also as a grouped launch.  Everything about the kernel is in
// conv_igemm_tile.h / conv_mfma.hip; this file only instantiates those templates in a translation unit of their own (built beside
// conv_mfma.hip: each of these kernels carries the row loops of its epilogue specialised per switch combination).
#include "conv_igemm_tile.h"

namespace ron {
namespace detail {

int launch_igemm4w(int dtype, int bn, const ConvArgs& a, hipStream_t s) {
  if (bn == 256) {
    if (dtype == RON_DTYPE_BF16) return launch_t<TraitsBF16S, 256, 256, 2, 2, 2, 1>(a, s);
    if (dtype == RON_DTYPE_F16) return launch_t<TraitsF16S, 256, 256, 2, 2, 2, 1>(a, s);
    if (dtype == RON_DTYPE_F16X3) return launch_t<TraitsF16X3S, 256, 256, 2, 2, 2, 1>(a, s);
  } else if (bn == 128) {
    if (dtype == RON_DTYPE_BF16) return launch_t<TraitsBF16S, 256, 128, 2, 2, 2, 1>(a, s);
    if (dtype == RON_DTYPE_F16) return launch_t<TraitsF16S, 256, 128, 2, 2, 2, 1>(a, s);
    if (dtype == RON_DTYPE_F16X3) return launch_t<TraitsF16X3S, 256, 128, 2, 2, 2, 1>(a, s);
  }
  ron::set_error("conv: no four-wave tile 256 x %d for dtype %d", bn, dtype);
  return RON_ERR_INVALID;
}

int launch_igemm4w_group(int dtype, const ConvGroupArgs& g, bool any_split, hipStream_t s) {
  if (dtype == RON_DTYPE_BF16) return launch_group_t<TraitsBF16S, 256, 256, 2, 2, 2, 1>(g, any_split, s);
  if (dtype == RON_DTYPE_F16) return launch_group_t<TraitsF16S, 256, 256, 2, 2, 2, 1>(g, any_split, s);
  if (dtype == RON_DTYPE_F16X3) return launch_group_t<TraitsF16X3S, 256, 256, 2, 2, 2, 1>(g, any_split, s);
  ron::set_error("conv group: no four-wave 256 x 256 tile for dtype %d", dtype);
  return RON_ERR_INVALID;
}

}  // namespace detail
}  // namespace ron
