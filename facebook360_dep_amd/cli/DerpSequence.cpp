// DerpSequence — the depth_estimation stage of the reference's render pipeline
// (scripts/render/pipeline.py:364-408) as ONE native program per GPU: for every level, coarse to fine,
// DerpCLI(level) on every frame -> TemporalBilateralFilter(level) over [t - R, t + R] -> "Transfer" (the
// filtered level overwrites disparity_levels/level_L) -> next level. The reference runs those three steps as
// separate worker jobs that hand frames over through the file system; here the frames of a chunk stay
// resident in HBM (frame slots) — or, with --resident_frames N, stream level by level through N slots from host
// memory, so that the sequence length is bounded by host RAM, not by HBM — and with several GPUs each process
// owns a contiguous chunk of the frames (render.py:169-175) and exchanges only the halo frames' raw level
// disparity over RCCL (derp_seq_*, include/derp_hip.h) — C++ host + HIP + RCCL, no Python in the loop.
//
// I/O schedule: decoding starts before the HIP runtime does. The worker pool inflates every frame's coarse
// levels first (a quarter of the bytes), then the finest level frame by frame; the level loop consumes a
// (frame, level) as soon as it is decoded, so the finest level's PNGs inflate behind the coarse levels' compute
// and frame k + 1's behind frame k's. Results are written by the pool behind the next level.
//
// Flags: DerpCLI's (DerpCLI.cpp:40-67) + TemporalBilateralFilter's filter flags (:51-59) + the pipeline's
// do_temporal_filter / do_temporal_masking (pipeline.py:378,386). Inputs and outputs are the files the three
// reference binaries would read and leave behind: disparity_levels/level_L/<cam>/<frame>.pfm (filtered) and
// disparity_time_filtered_levels/level_L/<cam>/<frame>.pfm.
//
// Multi-GPU: `--gpus N` forks one process per GPU of this node; or start the ranks yourself with
// RANK / WORLD_SIZE / LOCAL_RANK in the environment (torchrun-style) and --rccl_id_file on a shared path.
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include "derp_job.h"
#include "rendezvous.h"

using namespace cli;

static const char* kUsage = R"(
  - Computes temporally filtered disparity maps for a sequence of frames: per pyramid level, depth
    estimation of every frame, the temporal joint bilateral filter, and the write-back that seeds the
    next level (the depth_estimation stage of scripts/render/pipeline.py in one program per GPU).

  - Example:
    ./DerpSequence \
    --input_root=/path/to/ \
    --output_root=/path/to/output \
    --first=000000 \
    --last=000007 \
    --gpus=8
)";

// ---- the "files" transport: the plan's transfers through <output_root>/.halo (one file per (level, kind, frame,
// receiver), written under a temporary name and renamed, removed by its reader) — the reference moves the window
// frames of TemporalBilateralFilter the same way, through the shared file system (pipeline.py:364-408)
struct FileTransport {
  derp_seq* seq;
  derp_ctx* ctx;
  int rank, world, first, last, radius, partition;
  fs::path dir;
  std::string token;  // the launch's rendezvous token: every halo file starts with it
  std::vector<derp_seq_transfer> plan;
  std::vector<char> buf;
  double seconds = 0;
  uint64_t received = 0;
  void init() {
    const int n = derp_seq_plan(first, last, world, radius, partition, nullptr, 0);
    plan.resize(std::max(n, 0));
    if (n > 0) {
      derp_seq_plan(first, last, world, radius, partition, plan.data(), n);
    }
    fs::create_directories(dir);
  }
  fs::path name(int level, int kind, const derp_seq_transfer& t) const {
    // the launch's token is part of the name as well as of the content: a file some other launch left here is never
    // the file this rank waits for
    return dir / fmt("L%d_k%d_f%06d_to%d.%s.bin", level, kind, t.frame, t.to_rank, token.c_str());
  }
  void move(int level, int kind) {
    Timer tm;
    for (const derp_seq_transfer& t : plan) {
      if (t.from_rank != rank) {
        continue;
      }
      void* p;
      size_t bytes;
      DERP_OK(ctx, derp_seq_buffer(seq, t.frame, level, kind, &p, &bytes));
      buf.resize(bytes);
      DERP_OK(ctx, derp_seq_buffer_copy(seq, t.frame, level, kind, buf.data(), bytes, 0));
      const fs::path dst = name(level, kind, t), tmp = dst.string() + ".tmp";
      {
        std::ofstream f(tmp, std::ios::binary);
        f.write(token.data(), (std::streamsize)token.size());
        f.write(buf.data(), (std::streamsize)bytes);
        CHECK_MSG(f.good(), "failed to write " + tmp.string());
      }
      fs::rename(tmp, dst);
    }
    for (const derp_seq_transfer& t : plan) {
      if (t.to_rank != rank) {
        continue;
      }
      void* p;
      size_t bytes;
      DERP_OK(ctx, derp_seq_buffer(seq, t.frame, level, kind, &p, &bytes));
      const fs::path src = name(level, kind, t);
      Timer w;
      std::error_code ec;
      while (!fs::exists(src, ec)) {
        CHECK_MSG(w.s() < 600, "timed out waiting for " + src.string());
        usleep(2000);
      }
      buf.resize(bytes);
      {
        std::ifstream f(src, std::ios::binary);
        std::string head(token.size(), '\0');
        f.read(&head[0], (std::streamsize)head.size());
        CHECK_MSG(f.gcount() == (std::streamsize)head.size() && head == token,
                  "halo file of another launch (stale?): " + src.string());
        f.read(buf.data(), (std::streamsize)bytes);
        CHECK_MSG(f.gcount() == (std::streamsize)bytes, "short read from " + src.string());
      }
      fs::remove(src, ec);
      DERP_OK(ctx, derp_seq_buffer_copy(seq, t.frame, level, kind, buf.data(), bytes, 1));
      received += bytes;
    }
    seconds += tm.s();
  }
};

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

int main(int argc, char** argv) {
  Flags F;
  F.usage_msg = kUsage;
  define_derp_flags(F);
  // TemporalBilateralFilter.cpp:51-59 + pipeline.py:378,386
  F.boolean("do_temporal_filter", true, "apply the temporal filter at each level [extension: pipeline.py do_temporal_filter]");
  F.boolean("do_temporal_masking", false, "use foreground masks in the temporal filter [extension: pipeline.py do_temporal_masking]");
  F.dbl("sigma", 0.01, "spatio-temporal smoothing [extension: TemporalBilateralFilter --sigma]");
  F.i32("space_radius", -1, "space filtering radius [extension: TemporalBilateralFilter --space_radius]");
  F.i32("time_radius", 2, "temporal filtering radius [extension: TemporalBilateralFilter --time_radius]");
  F.dbl("weight_b", 0.5, "Blue channel weight [extension: TemporalBilateralFilter --weight_b]");
  F.dbl("weight_g", 1.0, "Green channel weight [extension: TemporalBilateralFilter --weight_g]");
  F.dbl("weight_r", 1.0, "Red channel weight [extension: TemporalBilateralFilter --weight_r]");
  F.i32("gpus", 1, "fork one process per GPU of this node (1 = this process only) [extension]");
  F.str("partition", "block", "frames per rank: block (contiguous chunks) | cyclic [extension]");
  F.str("rccl_id_file", "", "file through which rank 0 hands the RCCL unique id to the other ranks [extension]");
  F.str("exchange", "rccl",
        "how ranks move the halo frames: rccl (send / recv over xGMI) | files (through --output_root, the way the "
        "reference's workers hand frames over; also what rccl falls back to when the communicator cannot be built, "
        "e.g. two ranks on one GPU) [extension]");
  F.i32("resident_frames", 0,
        "frames kept in HBM per GPU: 0 = all of them; N >= 2 * time_radius + 1 = out of core, the other frames stream "
        "level by level from host memory [extension]");
  F.parse(argc, argv);
  Timer total;
  // RCCL's peer-memory handles need the dmabuf IPC mode on this driver stack; keep the caller's choice if any
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);

  // ---- ranks: --gpus forks them; otherwise RANK / WORLD_SIZE / LOCAL_RANK from the environment
  int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), localRank = env_int("LOCAL_RANK", -1);
  std::string idFile = F.s("rccl_id_file");
  if (F.i("gpus") > 1 && world == 1) {
    world = F.i("gpus");
    CHECK_MSG(F.s("output_root") != "", "output_root");
    fs::create_directories(F.s("output_root"));
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {  // fork before any HIP call
      const pid_t pid = fork();
      CHECK_MSG(pid >= 0, "fork failed");
      if (pid == 0) {
        rank = r;
        localRank = r;
        kids.clear();
        break;
      }
      kids.push_back(pid);
    }
    if (!kids.empty()) {  // the parent only waits: a failing rank fails the job and takes the others down with it
      int failed = 0;        // (a rank that died would leave its peers blocked in the RCCL rendezvous / exchange)
      while (!kids.empty()) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, 0);
        if (done < 0) {
          break;
        }
        kids.erase(std::remove(kids.begin(), kids.end(), done), kids.end());  // reaped: its pid may be reused
        if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
          ++failed;
          for (pid_t k : kids) {  // only ranks that are still our un-reaped children
            kill(k, SIGTERM);
          }
        }
      }
      {
        std::error_code ec;
        fs::remove_all(fs::path(F.s("output_root")) / ".derp_seq", ec);  // rendezvous files (and, after a failure, halo files)
      }
      if (failed) {
        LOG_FATAL(fmt("%d of %d ranks failed", failed, world));
      }
      LOG_INFO(fmt("-- TOTAL: %.3fs wall on %d GPUs", total.s(), world));
      return EXIT_SUCCESS;
    }
  }
  CHECK_MSG(rank >= 0 && rank < world, "RANK < WORLD_SIZE");
  if (world > 1 && !F.s("log_dir").empty() && log_file()) {  // one log file per rank
    fclose(log_file());
    log_file() = fopen((fs::path(F.s("log_dir")) / fmt("%s.rank%d.INFO", F.program.c_str(), rank)).c_str(), "w");
  }

  DerpJob J(F);
  J.setup_host();
  CHECK_MSG(F.s("partition") == "block" || F.s("partition") == "cyclic", "partition is block or cyclic");
  CHECK_MSG(F.i("time_radius") >= 0, "time_radius >= 0");
  const int partition = F.s("partition") == "block" ? DERP_SEQ_BLOCK : DERP_SEQ_CYCLIC;
  const int first = J.firstFrame, last = J.firstFrame + J.numFrames - 1;
  std::vector<int> owned;
  for (int t = first; t <= last; ++t) {
    if (derp_seq_owner(first, last, world, partition, t) == rank) {
      owned.push_back(t);
    }
  }
  const int nOwned = (int)owned.size();

  // ---- decoding starts now, before the HIP runtime: coarse levels of every frame first, then the finest level
  // frame by frame (the order the level loop consumes them in)
  IoPool pool(F.i("threads"));
  FrameStore store(J, pool, owned);
  // (with --resident_frames the library streams from the decoded buffers for the whole run: no throttling then)
  store.throttle = !(F.i("resident_frames") > 0 && F.i("resident_frames") < nOwned);
  for (int level = store.inTop; level > J.levelEnd; --level) {
    for (int k = 0; k < nOwned; ++k) {
      store.schedule(k, level);
    }
  }
  for (int k = 0; k < nOwned; ++k) {
    store.schedule(k, J.levelEnd);
  }
  store.pump();

  const double tHost = total.s();
  // DERP_SINGLE_DEVICE: every rank on --device (several ranks sharing one GPU: tests on a one-GPU box)
  J.setup_device(world > 1 && !getenv("DERP_SINGLE_DEVICE") ? (localRank >= 0 ? localRank : rank) : -1);
  const double tDevice = total.s();
  J.create_output_dirs({"disparity_time_filtered_levels"});
  derp_ctx* ctx = J.ctx;

  derp_seq_options so;
  derp_seq_options_default(&so);
  so.time_radius = F.i("time_radius");
  so.sigma = (float)F.d("sigma");
  so.weight_b = (float)F.d("weight_b");
  so.weight_g = (float)F.d("weight_g");
  so.weight_r = (float)F.d("weight_r");
  so.space_radius = F.i("space_radius");
  so.use_foreground_masks = F.b("do_temporal_masking");
  so.do_temporal_filter = F.b("do_temporal_filter");
  so.partition = partition;
  so.resident_frames = F.i("resident_frames");
  CHECK_MSG(!so.use_foreground_masks || J.useFg, "do_temporal_masking needs --use_foreground_masks (the masks must be loaded)");
  derp_seq* seq = nullptr;
  DERP_OK(ctx, derp_seq_create(&seq, ctx, first, last, rank, world, &so));
  int nHalo = 0, nSlots = 0;
  derp_seq_counts(seq, nullptr, &nHalo);
  derp_frame_slots(ctx, &nSlots, nullptr);
  LOG_INFO(fmt("rank %d of %d: %d frame(s) owned, %d halo frame(s), %d frame slot(s) in HBM%s", rank, world, nOwned, nHalo,
               nSlots, nSlots < nOwned ? " (out of core)" : ""));

  CHECK_MSG(F.s("exchange") == "rccl" || F.s("exchange") == "files", "exchange is rccl or files");
  bool useFiles = world > 1 && F.s("exchange") == "files";
  Rendezvous rv;
  rv.dir = fs::path(J.outputRoot) / ".derp_seq";
  rv.rank = rank;
  rv.world = world;
  if (world > 1) {
    rv.join();  // a token of THIS launch on every rank; whatever an earlier job left under .derp_seq is gone
  }
  if (world > 1 && !useFiles) {  // RCCL communicator: rank 0 publishes the unique id through a file
    if (idFile.empty()) {
      idFile = (rv.dir / "rccl_id").string();
    }
    // Every step whose failure on ONE rank would leave the others blocked is agreed on first: can every rank load
    // librccl at all (ncclCommInitRank is a rendezvous: a rank that never calls it hangs the rest)?
    unsigned char id[128];
    const bool loadable = derp_rccl_unique_id(id, sizeof id) == 0;  // binds the library; rank 0 keeps this id
    const int canLoad = rv.agree("rccl_load", loadable);
    if (canLoad != 1) {
      LOG_WARNING(fmt("RCCL transport unavailable (librccl cannot be loaded on %s rank); exchanging the halo frames "
                      "through files", canLoad == 0 ? "any" : "some"));
      useFiles = true;
    } else {
      // file = [128-byte id][token]: only this launch's token makes it this launch's id
      if (rank == 0) {
        const std::string tmp = idFile + ".tmp";
        {
          std::ofstream f(tmp, std::ios::binary);
          f.write(reinterpret_cast<const char*>(id), sizeof id);
          f.write(rv.token.data(), (std::streamsize)rv.token.size());
        }
        fs::rename(tmp, idFile);
      } else {
        Timer t;
        for (;;) {
          std::error_code ec;
          if (fs::exists(idFile, ec) && fs::file_size(idFile, ec) == sizeof id + rv.token.size()) {
            std::ifstream f(idFile, std::ios::binary);
            std::string got(rv.token.size(), '\0');
            f.read(reinterpret_cast<char*>(id), sizeof id);
            f.read(&got[0], (std::streamsize)got.size());
            if (f && got == rv.token) {
              break;
            }
          }
          CHECK_MSG(t.s() < 300, "timed out waiting for " + idFile);
          usleep(20000);
        }
      }
      const bool attached = derp_seq_attach_rccl(seq, id, sizeof id) == 0 && derp_seq_selftest(seq, 4096) == 0;
      const std::string why = attached ? "" : derp_last_error(ctx);
      // all ranks use RCCL or all ranks use files: a rank that switched alone would leave its peers in a collective
      const int all = rv.agree("rccl_attach", attached);
      if (all == 1) {
        if (rank == 0) {
          std::error_code ec;
          fs::remove(idFile, ec);  // every rank has joined the communicator: nobody needs the id any more
        }
      } else {
        LOG_WARNING("RCCL transport unavailable (" + (why.empty() ? std::string("another rank could not attach") : why) +
                    "); exchanging the halo frames through files");
        useFiles = true;
      }
    }
  }
  FileTransport files{seq, ctx, rank, world, first, last, so.do_temporal_filter ? so.time_radius : 0, partition,
                      rv.dir / "halo", rv.token};
  if (useFiles) {
    DERP_OK(ctx, derp_seq_attach_external(seq));
    files.init();
    LOG_INFO("halo exchange through files under " + files.dir.string());
  }

  LOG_INFO(fmt("-- start-up: flags + rig + input check %.3fs, HIP runtime + context %.3fs, output dirs + frame slots + "
               "communicator %.3fs (images decoding since %.3fs)", tHost, tDevice - tHost, total.s() - tDevice, tHost));
  // ---- the level loop (pipeline.py:364-408)
  LevelWriter writer(J, pool);
  // the page-locked upload bounce planes, once, at the finest level's plane size
  if (nSlots >= nOwned) {
    store.reserve_bounce(J.levelEnd);
  }
  writer.reserve_ring(3);  // frames in flight between the GPU and the disk
  double tCompute = 0, tUpload = 0;
  const std::vector<fs::path> dirs = so.do_temporal_filter
      ? std::vector<fs::path>{J.dispLevels, fs::path(J.outputRoot) / "disparity_time_filtered_levels"}
      : std::vector<fs::path>{J.dispLevels};
  if (store.inTop > J.levelStart) {  // coarse masks / the previous level of a resumed run (DerpCLI.cpp:280-288)
    for (int k = 0; k < nOwned; ++k) {
      store.wait(k, store.inTop);
      store.hand_over(seq, k, store.inTop, nSlots >= nOwned);
    }
  }
  for (int level = J.levelStart; level >= J.levelEnd; --level) {
    LOG_INFO(fmt("Processing level %d of %d frame(s)", level, nOwned));
    const double lv0 = total.s(), dec0 = store.waited, up0 = tUpload, dl0 = writer.downloading, wr0 = writer.waited;
    // A frame is filtered as soon as its window is computed (derp_seq_level_filter_frame) and its files leave one
    // frame later, from the filter's scratch over the copy stream — behind the compute of the frames that follow,
    // instead of all at once after the level's last frame (at the finest level that was 4.3 GB of PFMs and most of a
    // second at the very end of the job). What the two directories receive is the same filtered level either way.
    // Per frame, not a running prefix: on a rank whose first owned frame reaches into another rank's chunk (every rank
    // but the first of a block partition, every rank of a cyclic one) that frame cannot be filtered before the level's
    // exchange, and the interior frames behind it still leave early.
    std::vector<char> filtered(nOwned, 0), saved(nOwned, 0);
    for (int k = 0; k < nOwned; ++k) {
      store.wait(k, level);  // this frame's level is decoded (the pool is busy with later frames / finer levels)
      Timer t;
      store.hand_over(seq, k, level, nSlots >= nOwned);
      tUpload += t.s();
      DERP_OK(ctx, derp_seq_level_compute_frame(seq, level, owned[k]));
      // (only at the two finest levels: above them a level's files are a few MB, and waiting for the GPU once per
      // frame instead of once per level costs more than they do)
      if (so.do_temporal_filter && level <= J.levelEnd + 1) {
        std::vector<int> ready;  // filtered in an earlier iteration: their kernels ran before this frame's compute
        for (int j = 0; j < nOwned; ++j) {
          if (filtered[j] && !saved[j]) {
            ready.push_back(j);
          }
        }
        for (int j = 0; j <= k; ++j) {  // a frame's window never reaches past frame j + radius: later ones cannot be ready
          if (!filtered[j]) {
            const int rc = derp_seq_level_filter_frame(seq, level, owned[j]);
            if (rc == 2) {
              continue;  // its window is not complete yet (a later frame, or a halo frame the exchange brings)
            }
            DERP_OK(ctx, rc);
            filtered[j] = 1;
          }
        }
        for (int j : ready) {
          writer.save_seq(seq, owned[j], level, zero_pad(owned[j]), dirs, level == J.levelEnd, true);
          saved[j] = 1;
        }
      }
    }
    {
      Timer t;
      if (nOwned == 0) {
        DERP_OK(ctx, derp_seq_level_compute(seq, level));
      }
      if (world > 1) {
        DERP_OK(ctx, derp_seq_exchange_inputs_level(seq, level));  // colour guides (+ masks) of the halo frames
        if (useFiles && so.do_temporal_filter) {
          files.move(level, 0);
          if (so.use_foreground_masks) {
            files.move(level, 1);
          }
        }
      }
      DERP_OK(ctx, derp_seq_level_exchange(seq, level));
      if (useFiles && so.do_temporal_filter) {
        files.move(level, 2);
        DERP_OK(ctx, derp_seq_mark_exchanged(seq, level));
      }
      DERP_OK(ctx, derp_seq_level_filter(seq, level));
      DERP_OK(ctx, derp_synchronize(ctx));
      tCompute += t.s();
    }
    for (int k = 0; k < nOwned; ++k) {
      if (!saved[k]) {
        // PNG only at the finest level: the pipeline forces PFM above it (pipeline.py:366-369)
        writer.save_seq(seq, owned[k], level, zero_pad(owned[k]), dirs, level == J.levelEnd);
      }
    }
    LOG_INFO(fmt("-- level %d: %.3fs (waited for decode %.3fs, input hand-over %.3fs, result downloads incl. waiting for "
                 "the GPU %.3fs, waited for a free download plane %.3fs)", level, total.s() - lv0, store.waited - dec0,
                 tUpload - up0, writer.downloading - dl0, writer.waited - wr0));
    LOG_INFO(fmt("-- Elapsed time: %.3fs wall (level %d)", total.s(), level));
  }
  {
    Timer t;
    writer.finish();
    LOG_INFO(fmt("-- waited %.3fs for the last files", t.s()));
  }
  if (useFiles) {
    LOG_INFO(fmt("-- rank %d: halo exchange through files: %.1f MB received, %.3fs", rank, files.received / 1e6, files.seconds));
  }
  if (world > 1) {
    rv.leave();
  }
  uint64_t sent = 0, received = 0;
  double exchangeMs = 0;
  derp_seq_stats(seq, &sent, &received, &exchangeMs);
  LOG_INFO(fmt("-- rank %d: waited for decode %.3fs, input hand-over %.3fs, exchange + filter %.3fs, halo exchange %.1f MB "
               "received / %.1f MB sent in %.1f ms on the stream, download %.3fs, waited for writes %.3fs",
               rank, store.waited, tUpload, tCompute, received / 1e6, sent / 1e6, exchangeMs, writer.downloading,
               writer.waited));
  char name[256];
  derp_device_name(ctx, name, sizeof name);
  LOG_INFO(fmt("-- TOTAL: %.3fs wall on %s (rank %d of %d)", total.s(), name, rank, world));
  {
    Timer t;
    derp_seq_destroy(seq);
    derp_destroy(ctx);
    LOG_INFO(fmt("-- released the device in %.3fs", t.s()));
  }
  return EXIT_SUCCESS;
}
