// TEST INFRASTRUCTURE: the product's DEFLATE decoder (uvc_amd/csrc/uvc_inflate_core.h, the body of the GPU thread) compiled for the host so
// that tests/test_inflate.py can run it against zlib without a GPU.  Built on the fly by the test; not part of any shipped library.
#include "../../uvc_amd/csrc/uvc_inflate_core.h"
extern "C" int inflate_core_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len) {
    static thread_local InflState S;
    return uvc_inflate_block(in, in_len, out, out_len, S);
}
