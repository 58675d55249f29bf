// stft_bluestein.hip -- arbitrary (non power-of-two) nfft via the chirp-z transform on top of
// power-of-two FFTs.  Placeholder: wired in a later step of this round.
#include "spectro_internal.h"

namespace sg {

int build_bluestein_tables(sg_plan& p) {
    set_error("nfft=%d is not a power of two (or exceeds the LDS budget); Bluestein path not built yet", p.nfft);
    return SG_ERR_UNSUPPORTED;
}

int launch_bluestein(const sg_plan& p, const StftArgs&) {
    set_error("nfft=%d: Bluestein path not built yet", p.nfft);
    return SG_ERR_UNSUPPORTED;
}

}  // namespace sg
