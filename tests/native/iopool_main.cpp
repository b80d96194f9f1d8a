// Test harness (CPU only) for cli_common.h's IoPool: <workers> long class-2 jobs per worker (each sleeps 200 ms), then
// one class-1 and one class-0 job. Prints how long those two waited for a worker, in ms, and how many class-2 jobs ran.
#include <atomic>
#include <cstdio>

#include "../../facebook360_dep_amd/cli/cli_common.h"

int main(int argc, char** argv) {
  const int workers = argc > 1 ? atoi(argv[1]) : 12;
  std::atomic<int> ran{0};
  double lat1 = -1, lat0 = -1;
  {
    cli::IoPool pool(workers);
    cli::IoBatch all;
    for (int i = 0; i < workers * 4; ++i) {
      all.add(pool, [&] {
        usleep(200000);
        ++ran;
      });
    }
    usleep(20000);  // every general worker is inside a long job now
    cli::Timer t;
    cli::IoBatch urgent;
    urgent.add(pool, [&] { lat1 = t.s() * 1e3; }, 1);
    urgent.add(pool, [&] { lat0 = t.s() * 1e3; }, 0);
    urgent.wait();
    all.wait();
  }
  printf("%.2f %.2f %d\n", lat1, lat0, ran.load());
  return 0;
}
