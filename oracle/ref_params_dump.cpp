// TEST INFRASTRUCTURE (oracle/): dumps the reference's own parameter defaults.
//
// Compiles the reference's CmdLineArgs.hpp *where it lies* (-I/root/reference; it only needs
// common.hpp and the CLI11 header vendored in the reference tree, no htslib), instantiates
// `CommandLineArgs` and prints, for every row of include/uvc_params.def, the reference's default.
// The output is committed as tests/golden/params_default.json; nothing of the reference is copied.
// Build + run: `make -C oracle ref_params` (only where /root/reference exists).
#include "CmdLineArgs.hpp"
#include <cstdio>

struct Dump : CommandLineArgs {
    // the two predicates the hot path evaluates on vcf_tumor_fname (common.hpp:56, main.hpp:2564)
    int tumor_vcf_is_provided = IS_PROVIDED(vcf_tumor_fname) ? 1 : 0;
    int tumor_vcf_fname_nonempty = (vcf_tumor_fname.size() > 0) ? 1 : 0;
};

int main() {
    Dump a;
    bool first = true;
    printf("{\n");
#define UVC_PI(n, d) printf("%s  \"%s\": %lld", first ? "" : ",\n", #n, (long long)(a.n)); first = false;
#define UVC_PD(n, d) printf("%s  \"%s\": %.17g", first ? "" : ",\n", #n, (double)(a.n)); first = false;
#include "uvc_params.def"
#undef UVC_PI
#undef UVC_PD
    // the family-assignment parameters (include/uvc_group_params.def), prefixed "group."
#define UVC_GI(n, d) printf(",\n  \"group.%s\": %lld", #n, (long long)(a.n));
#define UVC_GD(n, d) printf(",\n  \"group.%s\": %.17g", #n, (double)(a.n));
#include "uvc_group_params.def"
    printf("\n}\n");
    return 0;
}
