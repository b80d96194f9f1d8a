// DerpSequence — the depth_estimation stage of the reference's render pipeline
// (scripts/render/pipeline.py:364-408) as ONE native program per GPU: for every level, coarse to fine,
// DerpCLI(level) on every frame -> TemporalBilateralFilter(level) over [t - R, t + R] -> "Transfer" (the
// filtered level overwrites disparity_levels/level_L) -> next level. The reference runs those three steps as
// separate worker jobs that hand frames over through the file system; here the frames of a chunk stay
// resident in HBM (frame slots), and with several GPUs each process owns a contiguous chunk of the frames
// (render.py:169-175) and exchanges only the halo frames' raw level disparity over RCCL (derp_seq_*,
// include/derp_hip.h) — C++ host + HIP + RCCL, no Python in the loop.
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
    if (idFile.empty()) {
      idFile = (fs::path(F.s("output_root")) / fmt(".derp_rccl_id.%d", (int)getpid())).string();
    }
    fs::remove(idFile);
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
      size_t alive = kids.size();
      while (alive > 0) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, 0);
        if (done < 0) {
          break;
        }
        --alive;
        if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
          ++failed;
          for (pid_t k : kids) {
            if (k != done) {
              kill(k, SIGTERM);  // harmless for ranks that already exited
            }
          }
        }
      }
      fs::remove(idFile);
      if (failed) {
        LOG_FATAL(fmt("%d of %d ranks failed", failed, world));
      }
      LOG_INFO(fmt("-- TOTAL: %.3fs wall on %d GPUs", total.s(), world));
      return EXIT_SUCCESS;
    }
  }
  CHECK_MSG(rank >= 0 && rank < world, "RANK < WORLD_SIZE");

  DerpJob J(F);
  J.setup(world > 1 ? (localRank >= 0 ? localRank : rank) : -1);
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
  CHECK_MSG(F.s("partition") == "block" || F.s("partition") == "cyclic", "partition is block or cyclic");
  so.partition = F.s("partition") == "block" ? DERP_SEQ_BLOCK : DERP_SEQ_CYCLIC;
  CHECK_MSG(!so.use_foreground_masks || J.useFg, "do_temporal_masking needs --use_foreground_masks (the masks must be loaded)");
  const int first = J.firstFrame, last = J.firstFrame + J.numFrames - 1;
  derp_seq* seq = nullptr;
  DERP_OK(ctx, derp_seq_create(&seq, ctx, first, last, rank, world, &so));
  int nOwned = 0, nHalo = 0;
  derp_seq_counts(seq, &nOwned, &nHalo);
  std::vector<int> owned(std::max(nOwned, 1));
  derp_seq_frames(seq, 0, owned.data(), nOwned);
  owned.resize(nOwned);
  LOG_INFO(fmt("rank %d of %d: %d frame(s) owned, %d halo frame(s)", rank, world, nOwned, nHalo));

  if (world > 1) {  // RCCL communicator: rank 0 publishes the unique id through a file
    CHECK_MSG(!idFile.empty(), "--rccl_id_file (a path every rank can read) is needed when WORLD_SIZE > 1");
    unsigned char id[128];
    if (rank == 0) {
      CHECK_MSG(derp_rccl_unique_id(id, sizeof id) == 0, "ncclGetUniqueId failed (librccl not loadable?)");
      const std::string tmp = idFile + ".tmp";
      {
        std::ofstream f(tmp, std::ios::binary);
        f.write(reinterpret_cast<const char*>(id), sizeof id);
      }
      fs::rename(tmp, idFile);
    } else {
      Timer t;
      while (!fs::exists(idFile) || fs::file_size(idFile) < sizeof id) {
        CHECK_MSG(t.s() < 300, "timed out waiting for " + idFile);
        usleep(20000);
      }
      std::ifstream f(idFile, std::ios::binary);
      f.read(reinterpret_cast<char*>(id), sizeof id);
    }
    DERP_OK(ctx, derp_seq_attach_rccl(seq, id, sizeof id));
    DERP_OK(ctx, derp_seq_selftest(seq, 4096));
  }

  // ---- inputs: every owned frame into its slot (decode of frame k + 1 overlaps the upload of frame k)
  IoPool pool(F.i("threads"));
  {
    FrameStager stager(J, pool);
    if (nOwned) {
      stager.start_decode(owned[0], 0);
    }
    for (int k = 0; k < nOwned; ++k) {
      stager.wait(k & 1);
      if (k + 1 < nOwned) {
        stager.start_decode(owned[k + 1], (k & 1) ^ 1);
      }
      DERP_OK(ctx, derp_select_frame(ctx, derp_seq_frame_slot(seq, owned[k])));
      stager.upload(k & 1);
    }
    LOG_INFO(fmt("-- inputs of %d frame(s) resident in HBM after %.3fs (decode wait %.3fs, upload %.3fs)", nOwned,
                 total.s(), stager.waited, stager.uploading));
  }
  DERP_OK(ctx, derp_seq_exchange_inputs(seq));  // colour guides (+ masks) of the halo frames, once

  // ---- the level loop (pipeline.py:364-408)
  LevelWriter writer(J, pool);
  double tCompute = 0;
  const std::vector<fs::path> dirs = so.do_temporal_filter
      ? std::vector<fs::path>{J.dispLevels, fs::path(J.outputRoot) / "disparity_time_filtered_levels"}
      : std::vector<fs::path>{J.dispLevels};
  for (int level = J.levelStart; level >= J.levelEnd; --level) {
    LOG_INFO(fmt("Processing level %d of %d frame(s)", level, nOwned));
    {
      Timer t;
      DERP_OK(ctx, derp_seq_level_compute(seq, level));
      DERP_OK(ctx, derp_seq_level_exchange(seq, level));
      DERP_OK(ctx, derp_seq_level_filter(seq, level));
      DERP_OK(ctx, derp_synchronize(ctx));
      tCompute += t.s();
    }
    const int parity = level & 1;
    writer.begin(parity, J.npx(level) * 4 * J.D * std::max(nOwned, 1));
    for (int k = 0; k < nOwned; ++k) {
      DERP_OK(ctx, derp_select_frame(ctx, derp_seq_frame_slot(seq, owned[k])));
      writer.save(parity, J.npx(level) * 4 * J.D * k, level, zero_pad(owned[k]), dirs);
    }
    LOG_INFO(fmt("-- Elapsed time: %.3fs wall (level %d)", total.s(), level));
  }
  writer.finish();
  uint64_t sent = 0, received = 0;
  double exchangeMs = 0;
  derp_seq_stats(seq, &sent, &received, &exchangeMs);
  LOG_INFO(fmt("-- rank %d: compute + exchange + filter %.3fs, halo exchange %.1f MB received / %.1f MB sent in %.1f ms on "
               "the stream, download %.3fs, waited for writes %.3fs",
               rank, tCompute, received / 1e6, sent / 1e6, exchangeMs, writer.downloading, writer.waited));
  char name[256];
  derp_device_name(ctx, name, sizeof name);
  LOG_INFO(fmt("-- TOTAL: %.3fs wall on %s (rank %d of %d)", total.s(), name, rank, world));
  derp_seq_destroy(seq);
  derp_destroy(ctx);
  return EXIT_SUCCESS;
}
