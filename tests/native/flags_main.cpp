// Test harness (CPU only) for cli_common.h's gflags look-alike: declares one flag of every type the executables use,
// parses its own command line and prints "name=value" for each, one per line.
#include <cstdio>

#include "../../facebook360_dep_amd/cli/cli_common.h"

int main(int argc, char** argv) {
  cli::Flags F;
  F.usage_msg = "flags harness";
  F.str("input_root", "", "a string");
  F.str("cameras", "all", "another string");
  F.i32("threads", -1, "an int");
  F.i32("level_start", 9, "an int");
  F.dbl("sigma", 0.01, "a double");
  F.boolean("partial_coverage", false, "a bool");
  F.boolean("do_median_filter", true, "a bool that defaults to true");
  F.parse(argc, argv);
  printf("input_root=%s\ncameras=%s\nthreads=%d\nlevel_start=%d\nsigma=%.17g\npartial_coverage=%d\ndo_median_filter=%d\n",
         F.s("input_root").c_str(), F.s("cameras").c_str(), F.i("threads"), F.i("level_start"), F.d("sigma"),
         (int)F.b("partial_coverage"), (int)F.b("do_median_filter"));
  return 0;
}
