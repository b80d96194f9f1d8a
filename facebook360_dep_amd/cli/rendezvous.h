// Start-up rendezvous of the ranks of one DerpSequence launch through a directory on the shared file system.
#pragma once
#include <unistd.h>

#include "cli_common.h"

namespace cli {

// Start-up rendezvous of the ranks of ONE launch through <output_root>/.derp_seq — the only channel ranks that
// were started by hand (RANK / WORLD_SIZE) share. It gives every rank a token that belongs to this launch and
// nothing else, after rank 0 has wiped whatever an earlier (crashed) job left in the directory: round 3 named the
// halo files by MASTER_PORT:RUN_ID, which is often ":" or the same across runs, and a re-run on the same
// --output_root then read the dead job's disparities without any error.
//   rank 0: wipe the directory, publish a random token T, wait for ready.<T>.<r> of every r (each carries that
//           rank's own random word), publish go.<T> listing those words;
//   rank r: follow the token file (it may still be the dead job's, then it changes), answer every token seen with
//           ready.<T>.<r>, accept go.<T> only if it quotes the word this process wrote — a stale go cannot.
// agree(): every rank publishes ok / fail for a named step and reads everybody's: all ok, all failed, or mixed.
struct Rendezvous {
  fs::path dir;
  int rank = 0, world = 1;
  std::string token;
  static std::string random_word() {
    unsigned char b[12] = {0};
    if (FILE* f = fopen("/dev/urandom", "rb")) {
      const size_t got = fread(b, 1, sizeof b, f);
      (void)got;
      fclose(f);
    }
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::string w = fmt("%d-%lx-", (int)getpid(), (unsigned long)ts.tv_nsec);
    for (unsigned char c : b) {
      w += fmt("%02x", c);
    }
    return w;
  }
  static void publish(const fs::path& p, const std::string& text) {
    std::error_code ec;
    fs::create_directories(p.parent_path(), ec);
    const fs::path tmp = p.string() + fmt(".tmp%d", (int)getpid());
    {
      std::ofstream f(tmp, std::ios::binary);
      f << text;
    }
    fs::rename(tmp, p, ec);
  }
  static bool slurp(const fs::path& p, std::string& out) {
    std::ifstream f(p, std::ios::binary);
    if (!f) {
      return false;
    }
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
  }
  void join() {
    Timer t;
    if (rank == 0) {
      // The wipe must not race with the other ranks, which may already be answering the dead job's token inside the
      // directory (a file appearing under remove_all's feet makes it stop half way and leaves the rest behind): move
      // the old directory out of everybody's sight in one rename, then delete it at leisure.
      std::error_code ec;
      token = random_word();
      if (fs::exists(dir, ec)) {
        const fs::path grave = dir.string() + ".dead." + token;
        fs::rename(dir, grave, ec);
        const fs::path victim = ec ? dir : grave;  // the rename itself failed -> wipe in place
        for (int attempt = 0; attempt < 100; ++attempt) {
          std::error_code ec2;
          fs::remove_all(victim, ec2);
          if (!fs::exists(victim, ec2)) {
            break;
          }
        }
      }
      fs::create_directories(dir);
      publish(dir / "token", token);
      std::string go;
      for (int r = 1; r < world; ++r) {
        std::string word;
        while (!slurp(dir / fmt("ready.%s.%d", token.c_str(), r), word) || word.empty()) {
          CHECK_MSG(t.s() < 300, fmt("timed out waiting for rank %d at the start-up rendezvous in %s", r, dir.c_str()));
          usleep(5000);
        }
        go += word + "\n";
      }
      publish(dir / ("go." + token), go);
      return;
    }
    const std::string mine = random_word();
    std::string answered;
    for (;;) {
      std::string seen, go;
      if (slurp(dir / "token", seen) && !seen.empty()) {
        if (seen != answered) {
          publish(dir / fmt("ready.%s.%d", seen.c_str(), rank), mine);
          answered = seen;
        }
        if (slurp(dir / ("go." + seen), go) && go.find(mine + "\n") != std::string::npos) {
          token = seen;
          return;
        }
      }
      CHECK_MSG(t.s() < 300, "timed out at the start-up rendezvous in " + dir.string() + " (is rank 0 running?)");
      usleep(5000);
    }
  }
  // -> +1 every rank ok, 0 every rank failed, -1 mixed
  int agree(const std::string& step, bool ok) {
    publish(dir / fmt("%s.%s.%d", step.c_str(), token.c_str(), rank), ok ? "1" : "0");
    Timer t;
    int nOk = 0;
    for (int r = 0; r < world; ++r) {
      std::string v;
      while (!slurp(dir / fmt("%s.%s.%d", step.c_str(), token.c_str(), r), v) || v.empty()) {
        CHECK_MSG(t.s() < 600, fmt("timed out waiting for rank %d to report on '%s'", r, step.c_str()));
        usleep(5000);
      }
      nOk += v[0] == '1';
    }
    return nOk == world ? 1 : nOk == 0 ? 0 : -1;
  }
  // The directory goes when nobody reads it any more: every rank says "bye" after it has seen every "done", rank 0
  // removes it after every "bye" (removing it on "done" alone would take rank 0's own "done" from under a slower
  // rank that is still looking for it).
  void leave() {
    agree("done", true);
    if (rank != 0) {
      publish(dir / fmt("bye.%s.%d", token.c_str(), rank), "1");
      return;
    }
    Timer t;
    for (int r = 1; r < world; ++r) {
      std::string v;
      while (!slurp(dir / fmt("bye.%s.%d", token.c_str(), r), v) && t.s() < 60) {
        usleep(5000);
      }
    }
    std::error_code ec;
    fs::remove_all(dir, ec);
  }
};


}  // namespace cli
