// Test harness (CPU only) for the executables' input parsers in cli_common.h. usage: parser_main json|exr|pfm <file>
// Prints "ok <n cameras | w h>"; a file the parser refuses ends the process the way the executables end: a glog-style
// "F... Check failed" line on stderr, exit status 1 — never a crash.
#include <cstdio>

#include "../../facebook360_dep_amd/cli/cli_common.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    return 2;
  }
  const std::string kind = argv[1];
  int w = 0, h = 0;
  if (kind == "json") {
    printf("ok %zu\n", cli::load_rig(argv[2]).size());
  } else if (kind == "exr") {
    cli::read_exr_f32(argv[2], w, h);
    printf("ok %d %d\n", w, h);
  } else {
    cli::read_pfm(argv[2], w, h);
    printf("ok %d %d\n", w, h);
  }
  return 0;
}
