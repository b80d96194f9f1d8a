// Test harness (CPU only) for cli/rendezvous.h: one rank of a launch. usage: rendezvous_main <dir> <rank> <world>
// <delay_ms before joining> <ok: 1|0 for the agreed step> [<chatter_ms>]. Prints "<token> <agreement> <entries found under <dir>/halo
// right after the rendezvous>" and exits 0.
#include <cstdio>
#include <cstdlib>

#include "../../facebook360_dep_amd/cli/rendezvous.h"

int main(int argc, char** argv) {
  if (argc < 6) {
    return 2;
  }
  cli::Rendezvous rv;
  rv.dir = argv[1];
  rv.rank = atoi(argv[2]);
  rv.world = atoi(argv[3]);
  usleep(1000 * atoi(argv[4]));
  if (argc > 6 && rv.rank != 0) {  // chatter: what a rank answering the dead job's token does, over and over, for argv[6] ms
    cli::Timer t;
    while (t.s() * 1000 < atoi(argv[6])) {
      cli::Rendezvous::publish(rv.dir / cli::fmt("ready.dead-job.%d", rv.rank), "word");
    }
  }
  rv.join();
  int leftovers = 0;
  {
    std::error_code ec;
    for (cli::fs::directory_iterator it(rv.dir / "halo", ec), end; !ec && it != end; it.increment(ec)) {
      ++leftovers;
    }
  }
  const int all = rv.agree("step", atoi(argv[5]) != 0);
  printf("%s %d %d\n", rv.token.c_str(), all, leftovers);
  fflush(stdout);
  rv.leave();
  return 0;
}
