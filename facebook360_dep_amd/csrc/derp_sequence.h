// Sequence driver: the per-level barrier schedule of the reference's render pipeline
// (scripts/render/pipeline.py:364-408) for the frames of a sequence sharded over GPUs.
//
//   for level L, coarse -> fine:
//     DerpCLI(level L) on every frame                              -> derp_seq_level_compute
//     TemporalBilateralFilter(level L) over [t - R, t + R]         -> derp_seq_level_exchange (halo
//        frames' raw level disparity, point to point) + derp_seq_level_filter
//     "Transfer": the filtered level overwrites disparity_levels/L -> end of derp_seq_level_filter
//
// One process per GPU, one derp_seq per process. A rank owns a set of frames (block partition like
// render.py:169-175's frame chunks, or cyclic t mod G), each resident in a frame slot of the context.
// The only data that crosses ranks inside the level loop is the raw level-L disparity of the frames a
// neighbour's window reaches into (TemporalBilateralFilter.cpp:96-119,139-160); the colour guides and
// foreground masks of those halo frames are inputs and move once (derp_seq_exchange_inputs).
// The frame-outer order cannot be used with the temporal filter on: level L of frame t is seeded by the
// FILTERED level L+1, which needs raw level L+1 of frames t +- R, which ... — the dependency cone widens
// by R frames per level, so every frame must finish level L before any frame starts level L-1.
//
// Out of core (derp_seq_options.resident_frames = N < owned frames): only N frame slots live in HBM. The inputs of
// every owned frame stay in caller-owned host memory (derp_seq_host_inputs), the results in page-locked host
// buffers of the library, and a level runs as  compute every frame (stream in colour L + the filtered level L+1,
// processLevel, stream the raw level out)  ->  exchange  ->  filter every frame over a sliding window of slots.
// The reference reaches the same independence from sequence length by round-tripping every level through the
// file system (render.py:169-175, pipeline.py:120-171,364-408). Same kernels on the same data: results are
// bit-identical to the resident mode.
//
// Transports: RCCL ncclSend/ncclRecv on the context's own stream (librccl is dlopen-ed on first use;
// no torch in the loop), same-process loopback (several ranks emulated on one GPU: tests), or external
// (the caller moves the buffers derp_seq_buffer names: torch.distributed / gloo tests).
// Included by derp_capi.hip (one translation unit).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

// ---- host-only plan ---------------------------------------------------------------------------
void seq_window(int t, int first, int last, int radius, int* lo, int* hi) {
  // populateMinMaxFrame (TemporalBilateralFilter.cpp:96-119): the frames of [t - R, t + R] that exist,
  // i.e. the window clamped to the sequence (pipeline.py:344-362 ships exactly chunk +- R to a worker)
  *lo = std::max(first, t - radius);
  *hi = std::min(last, t + radius);
}

int seq_owner(int first, int last, int world, int policy, int frame) {
  const int F = last - first + 1, i = frame - first;
  if (i < 0 || i >= F || world < 1) {
    return -1;
  }
  if (policy == DERP_SEQ_CYCLIC) {
    return i % world;
  }
  // balanced contiguous chunks: the first F % world ranks own one frame more
  const int q = F / world, r = F % world;
  const int big = r * (q + 1);
  return i < big ? i / (q + 1) : r + (q ? (i - big) / q : 0);
}

std::vector<derp_seq_transfer> seq_plan(int first, int last, int world, int radius, int policy) {
  std::vector<derp_seq_transfer> out;
  for (int t = first; t <= last; ++t) {  // frame t goes to the owner of every frame whose window holds t
    const int from = seq_owner(first, last, world, policy, t);
    std::vector<char> sent(world, 0);
    int lo, hi;
    seq_window(t, first, last, radius, &lo, &hi);  // |u - t| <= R is symmetric: same range
    for (int u = lo; u <= hi; ++u) {
      const int to = seq_owner(first, last, world, policy, u);
      if (to != from) {
        sent[to] = 1;
      }
    }
    for (int to = 0; to < world; ++to) {  // ascending (frame, to): one global order for every rank
      if (sent[to]) {
        out.push_back({t, from, to});
      }
    }
  }
  return out;
}

// ---- RCCL, bound at run time ---------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi* rccl_api() {
  static RcclApi api;
  if (api.lib || !api.error.empty()) {
    return &api;
  }
  // by soname first: a process that already carries an RCCL (PyTorch-ROCm bundles one) must keep using it
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (api.lib) {
      break;
    }
  }
  if (!api.lib) {
    api.error = std::string("cannot load librccl: ") + dlerror();
    return &api;
  }
  auto sym = [&](const char* n) {
    void* p = dlsym(api.lib, n);
    if (!p && api.error.empty()) {
      api.error = std::string("librccl lacks ") + n;
    }
    return p;
  };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!api.error.empty()) {
    api.lib = nullptr;
  }
  return &api;
}

enum { SEQ_NONE = 0, SEQ_LOOPBACK = 1, SEQ_RCCL = 2, SEQ_EXTERNAL = 3 };

}  // namespace

struct derp_seq {
  derp_ctx* c = nullptr;
  int first = 0, last = 0, rank = 0, world = 1;
  derp_seq_options opt;
  std::vector<int> owned, halo;                 // frame numbers, ascending
  std::vector<derp_seq_transfer> plan;          // every rank's transfers, in one global order
  // halo frames' data: [halo index][level]
  std::vector<std::vector<DevBuf>> haloColor, haloFg, haloDisp;
  std::vector<DevBuf> filtered;                 // [owned index]: [D][n_level] of the level being filtered
  DevBuf fov, winMask;                          // fov [D][n]; window masks [2R+1][D][n] (temporal masking only)
  int transport = SEQ_NONE;
  std::vector<derp_seq*> peers;
  ncclComm_t comm = nullptr;
  unsigned long long bytesSent = 0, bytesRecv = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> exchangeSpans;
  double exchangeMs = 0;
  // The level's exchange runs on its own stream (round 6): behind an event on the compute stream (every owned frame's raw
  // level is complete), beside the filter of the owned frames whose windows hold no halo frame. The compute stream waits
  // for `exchanged` before it filters the other frames and before the Transfer overwrites a raw level a send may still read.
  hipStream_t xStream = nullptr;
  hipEvent_t computedEv = nullptr, exchangedEv = nullptr;
  bool exchangePending = false;                 // an exchange was issued on xStream and the compute stream has not waited for it
  std::vector<std::pair<hipEvent_t, hipEvent_t>> waitSpans;  // (compute stream reached its wait, exchange finished)
  double exchangeExposedMs = 0;                 // the part of exchangeMs the compute stream really waited for
  int levelReady = -1;                          // level whose compute finished and whose filter has not run yet
  int levelExchanged = -1;                      // level whose halo disparities have arrived (exchange ran / was marked)
  int computeLevel = -1, computedFrames = 0;    // progress of derp_seq_level_compute_frame over the owned frames
  std::vector<int> computedAt, filteredAt;      // [owned index]: the level whose raw result / filtered scratch the frame holds
  std::vector<uint64_t> uploaded;               // [owned index * numLevels + level]: destinations uploaded since the level last ran
  std::vector<hipEvent_t> filteredEv;           // [owned index]: recorded behind the frame's filter kernel
  int fovLevel = -1;                            // level q->fov was built for
  // ---- out-of-core mode (resident_frames < owned frames)
  bool streaming = false;
  int nSlots = 0;
  struct HostIn {
    const uint16_t* color = nullptr;            // [S][n][3] interleaved BGR u16 (what derp_upload_color takes)
    const uint8_t* fg = nullptr;                // [S][n]
    const float* bg = nullptr;                  // [D][n]
  };
  std::vector<std::vector<HostIn>> hostIn;      // [owned index][level], caller-owned memory
  std::vector<std::vector<float*>> hostDisp;    // [owned index][level]: page-locked [D][n], the level's RESULT
  std::vector<char> hostHave;                   // [owned index * numLevels + level]
  std::vector<float*> hostRaw;                  // [owned index]: page-locked [D][nmax], raw level between compute and filter
  struct SlotTag {
    int k = -1, colorLevel = -1, rawLevel = -1; // what a device slot holds right now
  };
  std::vector<SlotTag> slotTag;
  std::vector<int> edge;                        // owned frames some other rank's window reaches into (ascending)
  std::vector<std::vector<DevBuf>> edgeColor, edgeFg;  // [edge index][level]
  std::vector<DevBuf> edgeRaw;                  // [edge index]: raw level disparity [D][nmax]
  std::vector<char> edgeStaged;                 // [level]: the edge frames' inputs of the level are in their device buffers
};

namespace {

int owned_index(const derp_seq* q, int frame) {
  auto it = std::lower_bound(q->owned.begin(), q->owned.end(), frame);
  return it != q->owned.end() && *it == frame ? (int)(it - q->owned.begin()) : -1;
}
int halo_index(const derp_seq* q, int frame) {
  auto it = std::lower_bound(q->halo.begin(), q->halo.end(), frame);
  return it != q->halo.end() && *it == frame ? (int)(it - q->halo.begin()) : -1;
}
int edge_index(const derp_seq* q, int frame) {
  auto it = std::lower_bound(q->edge.begin(), q->edge.end(), frame);
  return it != q->edge.end() && *it == frame ? (int)(it - q->edge.begin()) : -1;
}

// pyramid buffers of a frame slot whether it is selected or parked
struct SlotView {
  std::vector<DevBuf>*color, *fg, *disp;
  std::vector<char>* haveDisp;
};
SlotView slot_view(derp_ctx* c, int slot) {
  if (slot == c->curSlot) {
    return {&c->pyrColor, &c->pyrFg, &c->pyrDisp, &c->haveDisp};
  }
  derp_ctx::FrameSlot& fs = c->parked[slot];
  return {&fs.pyrColor, &fs.pyrFg, &fs.pyrDisp, &fs.haveDisp};
}

// kind: 0 colour pyramid level (ushort4 [S][n]), 1 fg mask (u8 [S][n]), 2 level disparity (f32 [D][n])
int seq_buffer(derp_seq* q, int frame, int level, int kind, void** ptr, size_t* bytes) {
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  if (kind < 0 || kind > 2) {
    return fail(c, "unknown buffer kind %d", kind);
  }
  const size_t n = npx(c, level);
  const size_t sz = kind == 0 ? n * c->S * sizeof(ushort4) : kind == 1 ? n * c->S : n * c->D * sizeof(float);
  DevBuf* b = nullptr;
  const int oi = owned_index(q, frame);
  if (oi >= 0 && q->streaming) {
    // out of core: an owned frame has device buffers of its own only if another rank needs it
    const int ei = edge_index(q, frame);
    if (ei < 0) {
      return fail(c, "frame %d streams through the frame slots (resident_frames = %d) and is in no other rank's window: "
                     "it has no fixed device buffer", frame, q->nSlots);
    }
    b = kind == 0 ? &q->edgeColor[ei][level] : kind == 1 ? &q->edgeFg[ei][level] : &q->edgeRaw[ei];
    if (b->bytes < sz) {
      return fail(c, "edge buffer of frame %d level %d kind %d was not allocated", frame, level, kind);
    }
  } else if (oi >= 0) {
    SlotView v = slot_view(c, oi);
    b = kind == 0 ? &(*v.color)[level] : kind == 1 ? &(*v.fg)[level] : &(*v.disp)[level];
  } else {
    const int hi = halo_index(q, frame);
    if (hi < 0) {
      return fail(c, "frame %d is neither owned by rank %d nor in its temporal halo", frame, q->rank);
    }
    b = kind == 0 ? &q->haloColor[hi][level] : kind == 1 ? &q->haloFg[hi][level] : &q->haloDisp[hi][level];
    if (b->bytes < sz) {
      return fail(c, "halo buffer of frame %d level %d kind %d was not allocated", frame, level, kind);
    }
  }
  *ptr = b->p;
  *bytes = sz;
  return 0;
}

#define NCCLCHK(c, api, expr)                                                                    \
  do {                                                                                           \
    ncclResult_t r_ = (expr);                                                                    \
    if (r_ != ncclSuccess) {                                                                     \
      return fail(c, "RCCL error %s at %s:%d (%s)", (api)->GetErrorString(r_), __FILE__, __LINE__, #expr); \
    }                                                                                            \
  } while (0)

// this rank's sends and receives of one exchange over RCCL: one group on the exchange stream
int seq_exchange_rccl(derp_seq* q, int level, int kind) {
  derp_ctx* c = q->c;
  RcclApi* api = rccl_api();
  NCCLCHK(c, api, api->GroupStart());
  int rc = 0;
  for (const derp_seq_transfer& tr : q->plan) {
    void* p;
    size_t bytes;
    ncclResult_t r = ncclSuccess;
    if (tr.from_rank == q->rank) {
      rc = seq_buffer(q, tr.frame, level, kind, &p, &bytes);
      if (!rc) {
        r = api->Send(p, bytes, ncclUint8, tr.to_rank, q->comm, q->xStream);
        q->bytesSent += bytes;
      }
    } else if (tr.to_rank == q->rank) {
      rc = seq_buffer(q, tr.frame, level, kind, &p, &bytes);
      if (!rc) {
        r = api->Recv(p, bytes, ncclUint8, tr.from_rank, q->comm, q->xStream);
        q->bytesRecv += bytes;
      }
    }
    if (!rc && r != ncclSuccess) {
      rc = fail(c, "RCCL error %s moving frame %d (%d -> %d)", api->GetErrorString(r), tr.frame, tr.from_rank, tr.to_rank);
    }
    if (rc) {
      break;
    }
  }
  const ncclResult_t e = api->GroupEnd();  // always closed, also on the error path
  if (!rc && e != ncclSuccess) {
    rc = fail(c, "RCCL error %s at ncclGroupEnd", api->GetErrorString(e));
  }
  return rc;
}

int seq_stage_edges(derp_seq* q, int level);

// loopback: pull from the peer contexts of this process
int seq_exchange_loopback(derp_seq* q, int level, int kind) {
  derp_ctx* c = q->c;
  std::vector<char> synced(q->world, 0);
  for (const derp_seq_transfer& tr : q->plan) {
    if (tr.from_rank == q->rank) {
      void* p;
      size_t bytes;
      TRY(seq_buffer(q, tr.frame, level, kind, &p, &bytes));
      q->bytesSent += bytes;
    }
    if (tr.to_rank != q->rank) {
      continue;
    }
    derp_seq* peer = q->peers[tr.from_rank];
    if (!synced[tr.from_rank]) {
      if (kind != 2 && seq_stage_edges(peer, level)) {  // an out-of-core peer stages the inputs it is asked for
        return fail(c, "loopback peer %d: %s", tr.from_rank, peer->c->err.c_str());
      }
      HIPCHK(c, hipStreamSynchronize(peer->c->stream));
      synced[tr.from_rank] = 1;
    }
    void *src, *dst;
    size_t bs, bd;
    if (seq_buffer(peer, tr.frame, level, kind, &src, &bs)) {
      return fail(c, "loopback peer %d: %s", tr.from_rank, peer->c->err.c_str());
    }
    TRY(seq_buffer(q, tr.frame, level, kind, &dst, &bd));
    HIPCHK(c, hipMemcpyAsync(dst, src, bd, hipMemcpyDeviceToDevice, q->xStream));
    q->bytesRecv += bd;
  }
  HIPCHK(c, hipStreamSynchronize(q->xStream));  // pulls are complete before any peer overwrites its level
  return 0;
}

// move buffers of `kind` at `level` along the plan (this rank's sends and receives)
int seq_exchange(derp_seq* q, int level, int kind) {
  derp_ctx* c = q->c;
  if (q->plan.empty() || q->transport == SEQ_EXTERNAL) {
    return 0;
  }
  if (q->transport == SEQ_NONE) {
    return fail(c, "derp_seq: world size %d but no transport attached (rccl / loopback / external)", q->world);
  }
  hipEvent_t ea = nullptr, eb = nullptr;
  HIPCHK(c, hipEventCreate(&ea));
  if (hipEventCreate(&eb) != hipSuccess) {
    (void)hipEventDestroy(ea);
    return fail(c, "hipEventCreate failed");
  }
  // the exchange stream starts behind everything the compute stream holds so far (the owned frames' raw levels, or the
  // uploaded inputs) ...
  HIPCHK(c, hipEventRecord(q->computedEv, c->stream));
  HIPCHK(c, hipStreamWaitEvent(q->xStream, q->computedEv, 0));
  (void)hipEventRecord(ea, q->xStream);
  const int rc = q->transport == SEQ_RCCL ? seq_exchange_rccl(q, level, kind) : seq_exchange_loopback(q, level, kind);
  (void)hipEventRecord(eb, q->xStream);
  (void)hipEventRecord(q->exchangedEv, q->xStream);
  q->exchangeSpans.push_back({ea, eb});  // drained (and destroyed) by derp_seq_stats / derp_seq_destroy
  // ... and the compute stream waits for it (seq_join_exchange) at the first use of what it delivers
  q->exchangePending = true;
  return rc;
}

// the compute stream waits for the exchange in flight (no-op when there is none)
int seq_join_exchange(derp_seq* q) {
  derp_ctx* c = q->c;
  if (!q->exchangePending) {
    return 0;
  }
  hipEvent_t reached = nullptr, done = nullptr;
  if (hipEventCreate(&reached) == hipSuccess && hipEventCreate(&done) == hipSuccess) {
    (void)hipEventRecord(reached, c->stream);
  }
  HIPCHK(c, hipStreamWaitEvent(c->stream, q->exchangedEv, 0));
  if (reached && done) {
    (void)hipEventRecord(done, c->stream);  // right behind the wait: done - reached = what the compute stream stood still for
    q->waitSpans.push_back({reached, done});
  }
  q->exchangePending = false;
  return 0;
}

void seq_drain_spans(derp_seq* q) {
  for (auto& s : q->exchangeSpans) {
    (void)hipEventSynchronize(s.second);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s.first, s.second);
    q->exchangeMs += ms;
    (void)hipEventDestroy(s.first);
    (void)hipEventDestroy(s.second);
  }
  q->exchangeSpans.clear();
  for (auto& s : q->waitSpans) {
    (void)hipEventSynchronize(s.second);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s.first, s.second);
    q->exchangeExposedMs += ms > 0 ? ms : 0;
    (void)hipEventDestroy(s.first);
    (void)hipEventDestroy(s.second);
  }
  q->waitSpans.clear();
}

int temporal_space_radius(const derp_seq* q, int level) {
  // TemporalBilateralFilter.cpp:164-168: max(ceil(kTemporalSpaceRadiusMax * 0.9^level), kTemporalSpaceRadiusMin)
  if (q->opt.space_radius != -1) {
    return q->opt.space_radius;
  }
  const float scale = std::pow(0.9f, level);
  return (int)std::max(std::ceil(1 * scale), float(1));
}

}  // namespace

extern "C" {

void derp_seq_options_default(derp_seq_options* o) {
  o->time_radius = 2;         // --time_radius   TemporalBilateralFilter.cpp:54
  o->sigma = 0.01f;           // --sigma         :51
  o->weight_b = 0.5f;         // --weight_b      :57
  o->weight_g = 1.0f;         // --weight_g      :58
  o->weight_r = 1.0f;         // --weight_r      :59 (declared, never read: the call passes b, g, b — :176-178)
  o->space_radius = -1;       // --space_radius  :52
  o->use_foreground_masks = 0;  // pipeline.py:386 do_temporal_masking
  o->partition = DERP_SEQ_BLOCK;
  o->do_temporal_filter = 1;  // pipeline.py:378
  o->resident_frames = 0;     // every owned frame resident in HBM
}

void derp_seq_window(int frame, int first, int last, int time_radius, int* lo, int* hi) {
  int a, b;
  seq_window(frame, first, last, time_radius, &a, &b);
  if (lo) {
    *lo = a;
  }
  if (hi) {
    *hi = b;
  }
}

int derp_seq_owner(int first, int last, int world, int partition, int frame) {
  return seq_owner(first, last, world, partition, frame);
}

int derp_seq_plan(int first, int last, int world, int time_radius, int partition, derp_seq_transfer* out, int cap) {
  if (last < first || world < 1 || time_radius < 0) {
    return -1;
  }
  const std::vector<derp_seq_transfer> p = seq_plan(first, last, world, time_radius, partition);
  if (out) {
    for (int i = 0; i < (int)p.size() && i < cap; ++i) {
      out[i] = p[i];
    }
  }
  return (int)p.size();
}

int derp_rccl_unique_id(void* out, size_t cap) {
  RcclApi* api = rccl_api();
  if (!api->lib || !out || cap < sizeof(ncclUniqueId)) {
    return 1;
  }
  ncclUniqueId id;
  if (api->GetUniqueId(&id) != ncclSuccess) {
    return 1;
  }
  memcpy(out, &id, sizeof id);
  return 0;
}

int derp_seq_create(derp_seq** out, derp_ctx* c, int first, int last, int rank, int world, const derp_seq_options* opt) {
  if (!out || !c) {
    return 1;
  }
  *out = nullptr;
  if (c->numLevels == 0) {
    return fail(c, "derp_set_pyramid has not been called");
  }
  if (last < first || world < 1 || rank < 0 || rank >= world) {
    return fail(c, "bad sequence geometry: frames %d..%d, rank %d of %d", first, last, rank, world);
  }
  derp_seq* q = new derp_seq;
  q->c = c;
  q->first = first;
  q->last = last;
  q->rank = rank;
  q->world = world;
  if (opt) {
    q->opt = *opt;
  } else {
    derp_seq_options_default(&q->opt);
  }
  auto bail = [&](int rc) {
    derp_seq_destroy(q);
    return rc;
  };
  if (q->opt.time_radius < 0) {
    return bail(fail(c, "time_radius %d is negative", q->opt.time_radius));
  }
  if (!q->opt.do_temporal_filter) {
    q->opt.time_radius = 0;  // no window, no halo, no exchange: replicas
  }
  for (int t = first; t <= last; ++t) {
    if (seq_owner(first, last, world, q->opt.partition, t) == rank) {
      q->owned.push_back(t);
    }
  }
  q->plan = seq_plan(first, last, world, q->opt.time_radius, q->opt.partition);
  for (const derp_seq_transfer& tr : q->plan) {
    if (tr.to_rank == rank) {
      q->halo.push_back(tr.frame);  // plan is ordered by frame, one entry per (frame, to)
    }
  }
  if (hipSetDevice(c->device) != hipSuccess) {
    return bail(fail(c, "hipSetDevice failed"));
  }
  const int nOwned = (int)q->owned.size();
  q->streaming = q->opt.resident_frames > 0 && q->opt.resident_frames < nOwned;
  q->nSlots = q->streaming ? q->opt.resident_frames : std::max(1, nOwned);
  if (q->streaming && q->opt.do_temporal_filter && q->nSlots < 2 * q->opt.time_radius + 1) {
    return bail(fail(c, "resident_frames %d is smaller than the temporal window (2 * time_radius + 1 = %d frames)",
                     q->nSlots, 2 * q->opt.time_radius + 1));
  }
  if (derp_set_frame_slots(c, q->nSlots)) {
    return bail(1);
  }
  const int nl = c->numLevels;
  q->haloColor.assign(q->halo.size(), std::vector<DevBuf>(nl));
  q->haloFg.assign(q->halo.size(), std::vector<DevBuf>(nl));
  q->haloDisp.assign(q->halo.size(), std::vector<DevBuf>(nl));
  size_t nmax = 0;
  for (int l = 0; l < nl; ++l) {
    const size_t n = npx(c, l);
    nmax = std::max(nmax, n);
    if (n == 0) {
      continue;
    }
    for (size_t h = 0; h < q->halo.size(); ++h) {
      if (q->haloColor[h][l].ensure(n * c->S * sizeof(ushort4)) || q->haloDisp[h][l].ensure(n * c->D * sizeof(float)) ||
          (q->opt.use_foreground_masks && q->haloFg[h][l].ensure(n * c->S))) {
        return bail(fail(c, "out of device memory allocating the temporal halo (frame %d, level %d)", q->halo[h], l));
      }
    }
  }
  q->filtered.resize(q->streaming ? 1 : q->owned.size());  // out of core: one frame is filtered, then streamed out
  q->computedAt.assign(q->owned.size(), -1);
  q->filteredAt.assign(q->owned.size(), -1);
  if (hipStreamCreateWithFlags(&q->xStream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&q->computedEv, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&q->exchangedEv, hipEventDisableTiming) != hipSuccess) {
    return bail(fail(c, "hipStreamCreate / hipEventCreate failed"));
  }
  q->filteredEv.assign(q->streaming ? 0 : q->owned.size(), nullptr);
  for (auto& e : q->filteredEv) {
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      return bail(fail(c, "hipEventCreate failed"));
    }
  }
  for (auto& b : q->filtered) {
    if (b.ensure(nmax * c->D * sizeof(float))) {
      return bail(fail(c, "out of device memory allocating the filtered level"));
    }
  }
  if (q->streaming) {
    q->slotTag.assign(q->nSlots, derp_seq::SlotTag());
    q->hostIn.assign(nOwned, std::vector<derp_seq::HostIn>(nl));
    q->hostDisp.assign(nOwned, std::vector<float*>(nl, nullptr));
    q->hostHave.assign((size_t)nOwned * nl, 0);
    q->hostRaw.assign(nOwned, nullptr);
    for (int k = 0; k < nOwned; ++k) {
      if (q->opt.do_temporal_filter && hipHostMalloc((void**)&q->hostRaw[k], nmax * c->D * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        q->hostRaw[k] = nullptr;
        return bail(fail(c, "out of page-locked host memory for the raw level of frame %d", q->owned[k]));
      }
      for (int l = 0; l < nl; ++l) {
        const size_t n = npx(c, l);
        if (n && hipHostMalloc((void**)&q->hostDisp[k][l], n * c->D * sizeof(float), hipHostMallocDefault) != hipSuccess) {
          q->hostDisp[k][l] = nullptr;
          return bail(fail(c, "out of page-locked host memory for the results of frame %d", q->owned[k]));
        }
      }
    }
    for (const derp_seq_transfer& tr : q->plan) {  // frames this rank sends: they need fixed device buffers
      if (tr.from_rank == rank && (q->edge.empty() || q->edge.back() != tr.frame)) {
        q->edge.push_back(tr.frame);
      }
    }
    q->edgeColor.assign(q->edge.size(), std::vector<DevBuf>(nl));
    q->edgeFg.assign(q->edge.size(), std::vector<DevBuf>(nl));
    q->edgeRaw.resize(q->edge.size());
    for (size_t e = 0; e < q->edge.size(); ++e) {
      if (q->edgeRaw[e].ensure(nmax * c->D * sizeof(float))) {
        return bail(fail(c, "out of device memory allocating the edge frames"));
      }
      for (int l = 0; l < nl; ++l) {
        const size_t n = npx(c, l);
        if (n && (q->edgeColor[e][l].ensure(n * c->S * sizeof(ushort4)) ||
                  (q->opt.use_foreground_masks && q->edgeFg[e][l].ensure(n * c->S)))) {
          return bail(fail(c, "out of device memory allocating the edge frames"));
        }
      }
    }
  }
  if (q->fov.ensure(nmax * c->D) ||
      (q->opt.use_foreground_masks && q->winMask.ensure((size_t)(2 * q->opt.time_radius + 1) * nmax * c->D))) {
    return bail(fail(c, "out of device memory allocating the temporal masks"));
  }
  *out = q;
  return 0;
}

void derp_seq_destroy(derp_seq* q) {
  if (!q) {
    return;
  }
  if (q->c) {
    (void)hipSetDevice(q->c->device);
    (void)hipStreamSynchronize(q->c->stream);
  }
  seq_drain_spans(q);
  if (q->comm) {
    (void)rccl_api()->CommDestroy(q->comm);
  }
  for (auto* vv : {&q->haloColor, &q->haloFg, &q->haloDisp}) {
    for (auto& v : *vv) {
      for (auto& b : v) {
        b.release();
      }
    }
  }
  for (auto& b : q->filtered) {
    b.release();
  }
  for (auto& e : q->filteredEv) {
    if (e) {
      (void)hipEventDestroy(e);
    }
  }
  if (q->xStream) {
    (void)hipStreamSynchronize(q->xStream);
    (void)hipStreamDestroy(q->xStream);
  }
  for (hipEvent_t e : {q->computedEv, q->exchangedEv}) {
    if (e) {
      (void)hipEventDestroy(e);
    }
  }
  q->fov.release();
  q->winMask.release();
  for (auto& v : q->hostDisp) {
    for (float* p : v) {
      if (p) {
        (void)hipHostFree(p);
      }
    }
  }
  for (float* p : q->hostRaw) {
    if (p) {
      (void)hipHostFree(p);
    }
  }
  for (auto* vv : {&q->edgeColor, &q->edgeFg}) {
    for (auto& v : *vv) {
      for (auto& b : v) {
        b.release();
      }
    }
  }
  for (auto& b : q->edgeRaw) {
    b.release();
  }
  delete q;
}

int derp_seq_counts(const derp_seq* q, int* n_owned, int* n_halo) {
  if (!q) {
    return 1;
  }
  if (n_owned) {
    *n_owned = (int)q->owned.size();
  }
  if (n_halo) {
    *n_halo = (int)q->halo.size();
  }
  return 0;
}

int derp_seq_frames(const derp_seq* q, int halo, int* frames, int cap) {
  if (!q || !frames) {
    return 1;
  }
  const std::vector<int>& v = halo ? q->halo : q->owned;
  for (int i = 0; i < (int)v.size() && i < cap; ++i) {
    frames[i] = v[i];
  }
  return 0;
}

int derp_seq_frame_slot(const derp_seq* q, int frame) {
  return q ? owned_index(q, frame) : -1;
}

int derp_seq_buffer(derp_seq* q, int frame, int level, int kind, void** ptr, size_t* bytes) {
  if (!q || !ptr || !bytes) {
    return 1;
  }
  {
    const int jrc = seq_join_exchange(q);
    if (jrc) {
      return jrc;
    }
  }  // whoever asks for a buffer finds it behind the compute stream, like before round 6
  return seq_buffer(q, frame, level, kind, ptr, bytes);
}

int derp_seq_buffer_copy(derp_seq* q, int frame, int level, int kind, void* host, size_t bytes, int to_device) {
  if (!q || !host) {
    return 1;
  }
  derp_ctx* c = q->c;
  void* p;
  size_t sz;
  TRY(seq_buffer(q, frame, level, kind, &p, &sz));
  if (bytes != sz) {
    return fail(c, "buffer of frame %d level %d kind %d holds %zu bytes, not %zu", frame, level, kind, sz, bytes);
  }
  HIPCHK(c, hipSetDevice(c->device));
  TRY(seq_join_exchange(q));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the level's kernels wrote / will read it on the library's stream
  HIPCHK(c, hipMemcpy(to_device ? p : host, to_device ? host : p, sz, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
  return 0;
}

int derp_seq_attach_loopback(derp_seq* q, derp_seq* const* peers, int n) {
  if (!q || !peers || n != q->world) {
    return q ? fail(q->c, "loopback needs one derp_seq per rank (%d given, world %d)", n, q->world) : 1;
  }
  q->peers.assign(peers, peers + n);
  if (q->peers[q->rank] != q) {
    return fail(q->c, "peers[%d] must be this rank's own derp_seq", q->rank);
  }
  q->transport = SEQ_LOOPBACK;
  return 0;
}

int derp_seq_attach_external(derp_seq* q) {
  if (!q) {
    return 1;
  }
  q->transport = SEQ_EXTERNAL;
  return 0;
}

int derp_seq_attach_rccl(derp_seq* q, const void* unique_id, size_t bytes) {
  if (!q || !unique_id || bytes < sizeof(ncclUniqueId)) {
    return q ? fail(q->c, "derp_seq_attach_rccl needs the %zu-byte id from derp_rccl_unique_id", sizeof(ncclUniqueId)) : 1;
  }
  derp_ctx* c = q->c;
  RcclApi* api = rccl_api();
  if (!api->lib) {
    return fail(c, "%s", api->error.c_str());
  }
  HIPCHK(c, hipSetDevice(c->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  if (q->comm) {  // re-attached: the previous communicator is not leaked
    (void)api->CommDestroy(q->comm);
    q->comm = nullptr;
  }
  NCCLCHK(c, api, api->CommInitRank(&q->comm, q->world, id, q->rank));
  q->transport = SEQ_RCCL;
  return 0;
}

// one ring step over the attached transport: every rank sends `words` 32-bit words to rank + 1 and
// checks what rank - 1 sent (transport self-test; world 1 sends to itself inside one group)
int derp_seq_selftest(derp_seq* q, int words) {
  if (!q || words < 1) {
    return 1;
  }
  derp_ctx* c = q->c;
  if (q->transport != SEQ_RCCL) {
    return fail(c, "derp_seq_selftest exercises the RCCL transport; attach it first");
  }
  RcclApi* api = rccl_api();
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf a, b;
  int rc = 0;
  const size_t bytes = (size_t)words * 4;
  std::vector<uint32_t> h(words);
  const int next = (q->rank + 1) % q->world, prev = (q->rank + q->world - 1) % q->world;
  for (int i = 0; i < words; ++i) {
    h[i] = 0x9e3779b9u * (uint32_t)(i + 1) + (uint32_t)q->rank;
  }
  if (a.ensure(bytes) || b.ensure(bytes)) {
    rc = fail(c, "out of device memory");
  } else if (hipMemcpyAsync(a.p, h.data(), bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
             hipMemsetAsync(b.p, 0, bytes, c->stream) != hipSuccess) {
    rc = fail(c, "HIP error staging the self-test");
  } else {
    ncclResult_t r = api->GroupStart();
    if (r == ncclSuccess) {
      r = api->Send(a.p, bytes, ncclUint8, next, q->comm, c->stream);
    }
    if (r == ncclSuccess) {
      r = api->Recv(b.p, bytes, ncclUint8, prev, q->comm, c->stream);
    }
    const ncclResult_t e = api->GroupEnd();
    if (r == ncclSuccess) {
      r = e;
    }
    std::vector<uint32_t> got(words);
    if (r != ncclSuccess) {
      rc = fail(c, "RCCL self-test: %s", api->GetErrorString(r));
    } else if (hipStreamSynchronize(c->stream) != hipSuccess ||
               hipMemcpy(got.data(), b.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(c, "HIP error reading the self-test back: %s", hipGetErrorString(hipGetLastError()));
    } else {
      for (int i = 0; i < words && !rc; ++i) {
        if (got[i] != 0x9e3779b9u * (uint32_t)(i + 1) + (uint32_t)prev) {
          rc = fail(c, "RCCL self-test: word %d from rank %d is %08x", i, prev, got[i]);
        }
      }
    }
  }
  a.release();
  b.release();
  return rc;
}

// ---- out-of-core helpers ------------------------------------------------------------------------
}  // extern "C"
namespace {

// BGR u16 host plane -> BGRX texels of `dst` plane s (what derp_upload_color does for the selected slot)
int upload_color_plane(derp_ctx* c, DevBuf& dst, int s, const uint16_t* bgr, size_t n) {
  TRY(upload_tmp(c, c->staging, bgr, n * 3));
  hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->stream, c->staging.as<uint16_t>(),
                     dst.as<ushort4>() + (size_t)s * n, n);
  KCHECK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next plane
  return 0;
}

int host_inputs_of(derp_seq* q, int k, int level, const derp_seq::HostIn** out) {
  const derp_seq::HostIn& h = q->hostIn[k][level];
  if (!h.color && !h.fg) {
    return fail(q->c, "frame %d level %d: no host inputs registered (derp_seq_host_inputs)", q->owned[k], level);
  }
  *out = &h;
  return 0;
}

// out of core: inputs of the frames other ranks' windows reach into -> their fixed device buffers (once per level)
int seq_stage_edges(derp_seq* q, int level) {
  derp_ctx* c = q->c;
  if (!q->streaming || q->edge.empty()) {
    return 0;
  }
  if (q->edgeStaged.empty()) {
    q->edgeStaged.assign(c->numLevels, 0);
  }
  if (q->edgeStaged[level]) {
    return 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = npx(c, level);
  for (size_t e = 0; e < q->edge.size(); ++e) {
    const derp_seq::HostIn* h;
    TRY(host_inputs_of(q, owned_index(q, q->edge[e]), level, &h));
    if (!h->color) {
      return fail(c, "frame %d level %d: only masks were registered, the colour images are needed", q->edge[e], level);
    }
    for (int s = 0; s < c->S; ++s) {
      TRY(upload_color_plane(c, q->edgeColor[e][level], s, h->color + (size_t)s * n * 3, n));
    }
    if (q->opt.use_foreground_masks) {
      if (!h->fg) {
        return fail(c, "frame %d level %d: temporal masking needs the foreground masks", q->edge[e], level);
      }
      HIPCHK(c, hipMemcpy(q->edgeFg[e][level].p, h->fg, n * c->S, hipMemcpyHostToDevice));
    }
  }
  q->edgeStaged[level] = 1;
  return 0;
}

// colour (+ masks, background) of `level` of owned frame k into the SELECTED slot
int stream_in_level(derp_seq* q, int k, int level, bool forCompute) {
  derp_ctx* c = q->c;
  const derp_seq::HostIn* h;
  TRY(host_inputs_of(q, k, level, &h));
  if (!h->color) {
    return fail(c, "frame %d level %d: only masks were registered, the colour images are needed", q->owned[k], level);
  }
  const size_t n = npx(c, level);
  for (int s = 0; s < c->S; ++s) {
    TRY(upload_color_plane(c, c->pyrColor[level], s, h->color + (size_t)s * n * 3, n));
  }
  if (h->fg) {
    HIPCHK(c, hipMemcpy(c->pyrFg[level].p, h->fg, n * c->S, hipMemcpyHostToDevice));
  }
  if (forCompute && h->bg) {
    HIPCHK(c, hipMemcpy(c->pyrBg[level].p, h->bg, n * c->D * sizeof(float), hipMemcpyHostToDevice));
    c->haveBg[level] = 1;
  }
  return 0;
}

// processLevel(level) of owned frame k in out-of-core mode: stream in, compute, stream the raw level out
int stream_compute_frame(derp_seq* q, int level, int k) {
  derp_ctx* c = q->c;
  const int slot = k % q->nSlots;
  TRY(select_frame(c, slot));
  derp_seq::SlotTag& tag = q->slotTag[slot];
  if (tag.k != k || tag.colorLevel != level) {
    tag = derp_seq::SlotTag();
    TRY(stream_in_level(q, k, level, true));
  }
  const int up = level + 1;
  const bool seeded = up < c->numLevels && npx(c, up) != 0;
  if (seeded) {
    if (!q->hostHave[(size_t)k * c->numLevels + up]) {
      return fail(c, "Missing disparity of level %d needed to start level %d (frame %d)", up, level, q->owned[k]);
    }
    HIPCHK(c, hipMemcpy(c->pyrDisp[up].p, q->hostDisp[k][up], npx(c, up) * c->D * sizeof(float), hipMemcpyHostToDevice));
    c->haveDisp[up] = 1;
    if (c->opt.use_foreground_masks) {  // the masked upsample reads the coarse mask too (DerpCLI.cpp:280-285)
      const derp_seq::HostIn* hu;
      TRY(host_inputs_of(q, k, up, &hu));
      if (hu->fg) {
        HIPCHK(c, hipMemcpy(c->pyrFg[up].p, hu->fg, npx(c, up) * c->S, hipMemcpyHostToDevice));
      }
    }
  }
  TRY(process_level(c, level));
  const size_t bytes = npx(c, level) * c->D * sizeof(float);
  float* out = q->opt.do_temporal_filter ? q->hostRaw[k] : q->hostDisp[k][level];
  HIPCHK(c, hipMemcpyAsync(out, c->pyrDisp[level].p, bytes, hipMemcpyDeviceToHost, c->stream));
  const int ei = edge_index(q, q->owned[k]);
  if (ei >= 0) {
    HIPCHK(c, hipMemcpyAsync(q->edgeRaw[ei].p, c->pyrDisp[level].p, bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  tag.k = k;
  tag.colorLevel = tag.rawLevel = level;
  if (!q->opt.do_temporal_filter) {
    q->hostHave[(size_t)k * c->numLevels + level] = 1;
  }
  return 0;
}

// temporal filter of every owned frame in out-of-core mode: a sliding window of slots holds (colour L, raw L)
int stream_filter_level(derp_seq* q, int level, int W, int H, int radius) {
  derp_ctx* c = q->c;
  const size_t n = (size_t)W * H;
  for (int j = 0; j < (int)q->owned.size(); ++j) {
    const int t = q->owned[j];
    int lo, hi;
    seq_window(t, q->first, q->last, q->opt.time_radius, &lo, &hi);
    std::vector<const void*> wg(hi - lo + 1);
    std::vector<const float*> wi(hi - lo + 1);
    std::vector<const uint8_t*> wmk(hi - lo + 1);
    for (int u = lo; u <= hi; ++u) {
      const void *pc, *pd, *pm;
      const int ku = owned_index(q, u);
      if (ku >= 0) {
        const int slot = ku % q->nSlots;
        derp_seq::SlotTag& tag = q->slotTag[slot];
        if (tag.k != ku || tag.colorLevel != level || tag.rawLevel != level) {
          TRY(select_frame(c, slot));
          if (tag.k != ku || tag.colorLevel != level) {
            tag = derp_seq::SlotTag();
            TRY(stream_in_level(q, ku, level, false));
          }
          HIPCHK(c, hipMemcpy(c->pyrDisp[level].p, q->hostRaw[ku], n * c->D * sizeof(float), hipMemcpyHostToDevice));
          tag.k = ku;
          tag.colorLevel = tag.rawLevel = level;
        }
        SlotView v = slot_view(c, slot);
        pc = (*v.color)[level].p;
        pd = (*v.disp)[level].p;
        pm = (*v.fg)[level].p;
      } else {
        void* p;
        size_t b;
        TRY(seq_buffer(q, u, level, 0, &p, &b));
        pc = p;
        TRY(seq_buffer(q, u, level, 2, &p, &b));
        pd = p;
        pm = nullptr;
        if (q->opt.use_foreground_masks) {
          TRY(seq_buffer(q, u, level, 1, &p, &b));
          pm = p;
        }
      }
      if (q->opt.use_foreground_masks) {  // mask = fg & fov of each frame (TemporalBilateralFilter.cpp:150-160)
        uint8_t* wm = q->winMask.as<uint8_t>() + (size_t)(u - lo) * n * c->D;
        hipLaunchKernelGGL(k_and_masks, dim3(flat_grid(n), c->D), dim3(256), 0, c->stream, q->fov.as<uint8_t>(),
                           (const uint8_t*)pm, c->dst2src.as<int>(), 0, n, wm);
        KCHECK(c);
        pm = wm;
      } else {
        pm = q->fov.p;
      }
      wg[u - lo] = pc;
      wi[u - lo] = reinterpret_cast<const float*>(pd);
      wmk[u - lo] = reinterpret_cast<const uint8_t*>(pm);
    }
    TRY(temporal_launch(c, wg.data(), wi.data(), wmk.data(), hi - lo + 1, t - lo, W, H, c->D, q->opt.sigma, radius,
                        q->opt.weight_b, q->opt.weight_g, q->opt.weight_b, q->filtered[0].as<float>(), c->dst2src.as<int>()));
    // "Transfer": the filtered level is the frame's result; the raw levels stay untouched in hostRaw
    HIPCHK(c, hipMemcpyAsync(q->hostDisp[j][level], q->filtered[0].p, n * c->D * sizeof(float), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    q->hostHave[(size_t)j * c->numLevels + level] = 1;
  }
  return 0;
}

}  // namespace
extern "C" {

int derp_seq_host_inputs(derp_seq* q, int frame, int level, const uint16_t* color_bgr, const uint8_t* fg_masks,
                         const float* background_disparity) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0 || (!color_bgr && !fg_masks)) {  // masks alone: a level that only feeds the masked upsample
    return fail(c, k < 0 ? "frame %d is not owned by rank %d" : "frame %d: neither colour nor masks", frame, q->rank);
  }
  HIPCHK(c, hipSetDevice(c->device));
  if (q->streaming) {
    q->hostIn[k][level] = {color_bgr, fg_masks, background_disparity};
    if (!q->edgeStaged.empty() && edge_index(q, frame) >= 0) {
      q->edgeStaged[level] = 0;
    }
    for (derp_seq::SlotTag& tag : q->slotTag) {  // a slot holding an older version of these inputs is stale
      if (tag.k == k && tag.colorLevel == level) {
        tag = derp_seq::SlotTag();
      }
    }
    return 0;
  }
  const size_t n = npx(c, level);
  TRY(select_frame(c, k));
  for (int s = 0; color_bgr && s < c->S; ++s) {
    TRY(upload_color_plane(c, c->pyrColor[level], s, color_bgr + (size_t)s * n * 3, n));
  }
  if (fg_masks) {
    HIPCHK(c, hipMemcpy(c->pyrFg[level].p, fg_masks, n * c->S, hipMemcpyHostToDevice));
  }
  if (background_disparity) {
    HIPCHK(c, hipMemcpy(c->pyrBg[level].p, background_disparity, n * c->D * sizeof(float), hipMemcpyHostToDevice));
    c->haveBg[level] = 1;
  }
  return 0;
}

int derp_seq_upload_color_plane(derp_seq* q, int frame, int level, int s, const uint16_t* bgr) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0 || s < 0 || s >= c->S || !bgr) {
    return fail(c, "bad frame / source index / null image");
  }
  if (q->streaming) {
    return fail(c, "out of core the frames stream from the buffers given to derp_seq_host_inputs");
  }
  HIPCHK(c, hipSetDevice(c->device));
  // on the copy stream, into the frame's own slot: the frame computing on the main stream is another one
  const size_t n = npx(c, level);
  SlotView v = slot_view(c, k);
  ALLOC(c, c->copyStaging, n * 3 * sizeof(uint16_t));
  HIPCHK(c, hipMemcpyAsync(c->copyStaging.p, bgr, n * 3 * sizeof(uint16_t), hipMemcpyHostToDevice, c->copyStream));
  hipLaunchKernelGGL(k_bgr_to_bgrx, dim3(flat_grid(n)), dim3(256), 0, c->copyStream, c->copyStaging.as<uint16_t>(),
                     (*v.color)[level].as<ushort4>() + (size_t)s * n, n);
  KCHECK(c);
  HIPCHK(c, hipStreamSynchronize(c->copyStream));  // the staging buffer and `bgr` are free again
  return 0;
}

int derp_seq_upload_disparity(derp_seq* q, int frame, int level, int d, const float* disp) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0 || d < 0 || d >= c->D || !disp) {
    return fail(c, "bad frame / destination index / null image");
  }
  if (q->streaming) {
    memcpy(q->hostDisp[k][level] + (size_t)d * npx(c, level), disp, npx(c, level) * sizeof(float));
    q->hostHave[(size_t)k * c->numLevels + level] = 1;
    return 0;
  }
  TRY(select_frame(c, k));
  TRY(derp_upload_disparity(c, level, d, disp));
  if (q->uploaded.size() != q->owned.size() * (size_t)c->numLevels) {
    q->uploaded.assign(q->owned.size() * (size_t)c->numLevels, 0);
  }
  q->uploaded[(size_t)k * c->numLevels + level] |= 1ull << d;
  return 0;
}

int derp_seq_download_disparity(derp_seq* q, int frame, int level, int d, float* disp) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0 || d < -1 || d >= c->D || !disp) {
    return fail(c, "bad frame / destination index / null output");
  }
  const size_t n = npx(c, level);
  if (q->streaming) {
    if (!q->hostHave[(size_t)k * c->numLevels + level]) {
      return fail(c, "level %d of frame %d has not been processed", level, frame);
    }
    memcpy(disp, q->hostDisp[k][level] + (size_t)std::max(d, 0) * n, (d < 0 ? c->D : 1) * n * sizeof(float));
    return 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  TRY(select_frame(c, k));
  if (d < 0) {  // every destination's plane in one copy, [D][h*w]
    if (!c->haveDisp[level]) {
      return fail(c, "level %d has not been processed", level);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(disp, c->pyrDisp[level].p, n * c->D * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  return derp_download_disparity(c, level, d, disp);
}

int derp_seq_exchange_inputs_level(derp_seq* q, int level) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  HIPCHK(c, hipSetDevice(c->device));
  TRY(check_level(c, level));
  TRY(seq_stage_edges(q, level));  // out of core: the frames other ranks read, into their fixed device buffers
  TRY(seq_exchange(q, level, 0));
  if (q->opt.use_foreground_masks) {
    TRY(seq_exchange(q, level, 1));
  }
  return seq_join_exchange(q);  // inputs: once per sequence, outside every timed region — nothing to overlap with
}

int derp_seq_exchange_inputs(derp_seq* q) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  for (int l = 0; l < c->numLevels; ++l) {
    if (npx(c, l) == 0) {
      continue;
    }
    TRY(derp_seq_exchange_inputs_level(q, l));
  }
  return 0;
}

int derp_seq_level_compute_frame(derp_seq* q, int level, int frame) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  HIPCHK(c, hipSetDevice(c->device));
  TRY(seq_join_exchange(q));  // (a level's filter has joined already: only a caller that skipped it gets here pending)
  const int k = owned_index(q, frame);
  if (k < 0) {
    return fail(c, "frame %d is not owned by rank %d", frame, q->rank);
  }
  if (q->computeLevel != level) {
    q->computeLevel = level;
    q->computedFrames = 0;
    q->levelReady = -1;
  }
  // the projection warps depend on the rig and the level size only (precomputeProjections, Derp.cpp:955-976):
  // the first frame of the level builds them (always, when rebuild_warp_tables asks for the reference's
  // per-invocation rebuild), the other frames of the level reuse them
  const int rebuild = c->opt.rebuild_warp_tables;
  c->opt.rebuild_warp_tables = q->computedFrames == 0 ? rebuild : 0;
  int rc;
  if (q->streaming) {
    rc = stream_compute_frame(q, level, k);
  } else {
    rc = select_frame(c, k);
    if (!rc) {
      rc = process_level(c, level);
    }
  }
  c->opt.rebuild_warp_tables = rebuild;
  TRY(rc);
  q->computedAt[k] = level;
  q->filteredAt[k] = -1;
  if (++q->computedFrames >= (int)q->owned.size()) {
    q->levelReady = level;
  }
  return 0;
}

// An owned frame's raw level that was computed elsewhere and uploaded (derp_seq_upload_disparity): counts as this
// frame's compute of the level. It is what TemporalBilateralFilter's inputs are — DerpCLI's files of the level, read
// back from disk (TemporalBilateralFilter.cpp:139-160) — so that the executable can run on the sequence driver's
// resident frames and per-frame filter instead of re-reading every window frame for every frame it filters.
int derp_seq_level_provided_frame(derp_seq* q, int level, int frame) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0) {
    return fail(c, "frame %d is not owned by rank %d", frame, q->rank);
  }
  if (q->streaming) {
    return fail(c, "derp_seq_level_provided_frame needs the frames resident in HBM");
  }
  TRY(select_frame(c, k));
  // every destination must have been uploaded since this (frame, level) last counted as computed: haveDisp alone is set
  // by the first plane and survives from an earlier run of the level
  const uint64_t all = c->D >= 64 ? ~0ull : (1ull << c->D) - 1;
  const size_t slotLevel = (size_t)k * c->numLevels + level;
  if (!c->haveDisp[level] || q->uploaded.size() <= slotLevel || (q->uploaded[slotLevel] & all) != all) {
    return fail(c, "frame %d has no complete level %d disparity (derp_seq_upload_disparity of every destination first)", frame, level);
  }
  q->uploaded[slotLevel] = 0;
  if (q->computeLevel != level) {
    q->computeLevel = level;
    q->computedFrames = 0;
    q->levelReady = -1;
  }
  q->computedAt[k] = level;
  q->filteredAt[k] = -1;
  if (++q->computedFrames >= (int)q->owned.size()) {
    q->levelReady = level;
  }
  return 0;
}

// How many work lanes (derp_ctx::WorkLane) the frames of `level` run on: 0 = one after the other on the context's
// own working set. Coarse levels only (DERP_SEQ_LANE_MAX_WIDTH, default 256 px), frames resident, every destination's
// tables in one batch; DERP_SEQ_LANES (default 8) = 0 / 1 switches the lanes off.
int seq_lane_count(derp_seq* q, int level) {
  derp_ctx* c = q->c;
  const int maxLanes = getenv("DERP_SEQ_LANES") ? atoi(getenv("DERP_SEQ_LANES")) : 8;
  const int maxWidth = getenv("DERP_SEQ_LANE_MAX_WIDTH") ? atoi(getenv("DERP_SEQ_LANE_MAX_WIDTH")) : 256;
  if (q->streaming || maxLanes < 2 || (int)q->owned.size() < 2 || c->LW[level] > maxWidth || c->LH[level] > maxWidth ||
      getenv("DERP_TABLE_BUDGET_GB")) {
    return 0;
  }
  return std::min(maxLanes, (int)q->owned.size()) - 1;  // the first frame's lane is the context itself
}

int derp_seq_level_compute(derp_seq* q, int level) {
  if (!q) {
    return 1;
  }
  q->computeLevel = -1;  // a fresh pass over the level
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  TRY(seq_join_exchange(q));
  const int lanes = seq_lane_count(q, level);
  if (lanes > 0) {
    // The first owned frame runs on the context's own stream and leaves the level's shared tables behind (projection
    // warps, pixel rays, resampling tables); the other frames follow on the work lanes, behind an event recorded at its
    // end, and overlap each other. The context's stream then waits for every lane: the exchange and the filter of the
    // level are ordered behind all frames, as before. Same kernels on the same data as the sequential order.
    HIPCHK(c, hipSetDevice(c->device));
    TRY(lanes_prepare(c, lanes, npx(c, level)));
    // the per-stage spans of the frames overlap in time on the lanes: this span on the context's stream — from the first
    // frame's first kernel to the join — is the level's wall ("lanes_wall")
    Span wall(c, ST_LANES, level);
    for (int k = 0; k < (int)q->owned.size(); ++k) {
      const int lane = k == 0 ? -1 : (k - 1) % lanes;
      if (q->computeLevel != level) {
        q->computeLevel = level;
        q->computedFrames = 0;
        q->levelReady = -1;
      }
      const int rebuild = c->opt.rebuild_warp_tables;
      c->opt.rebuild_warp_tables = k == 0 ? rebuild : 0;
      const int rc = process_level_on_lane(c, lane, k, level);
      c->opt.rebuild_warp_tables = rebuild;
      TRY(rc);
      if (k == 0) {
        if (c->DB != c->D) {
          return fail(c, "work lanes need every destination's tables in one batch (level %d: %d of %d)", level, c->DB, c->D);
        }
        HIPCHK(c, hipEventRecord(c->laneReady, c->stream));
      }
      q->computedAt[k] = level;
      q->filteredAt[k] = -1;
      ++q->computedFrames;
    }
    for (int i = 0; i < lanes; ++i) {
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->lanes[i]->done, 0));
    }
    q->levelReady = level;
    return 0;
  }
  for (int t : q->owned) {
    TRY(derp_seq_level_compute_frame(q, level, t));
  }
  if (q->owned.empty()) {
    q->levelReady = level;
  }
  return 0;
}

int derp_seq_level_exchange(derp_seq* q, int level) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  HIPCHK(c, hipSetDevice(c->device));
  if (!q->opt.do_temporal_filter) {
    return 0;
  }
  if (q->levelReady != level) {
    return fail(c, "derp_seq_level_exchange(%d): derp_seq_level_compute(%d) has not completed for every owned frame", level, level);
  }
  TRY(seq_exchange(q, level, 2));
  if (q->transport != SEQ_EXTERNAL) {
    q->levelExchanged = level;
  }
  if (q->streaming || getenv("DERP_SEQ_NO_OVERLAP")) {  // (developer A/B: the exchange in line with the compute stream)
    TRY(seq_join_exchange(q));
  }
  return 0;
}

int derp_seq_mark_exchanged(derp_seq* q, int level) {
  if (!q) {
    return 1;
  }
  q->levelExchanged = level;
  return 0;
}

namespace {
// q->fov = the destinations' FOV masks at this level (generateFovMasks), built once per level
int seq_fov_masks(derp_seq* q, int level) {
  derp_ctx* c = q->c;
  if (q->fovLevel != level) {
    const int W = c->LW[level], H = c->LH[level];
    hipLaunchKernelGGL(k_fov_mask, grid2d(W, H, c->D, kBlk2d), kBlk2d, 0, c->stream, c->camsDst.as<Cam>(), W, H,
                       q->fov.as<uint8_t>());
    KCHECK(c);
    q->fovLevel = level;
  }
  return 0;
}
// temporalJointBilateralFilter of owned frame k over its window into q->filtered[k] (resident mode)
int seq_filter_frame(derp_seq* q, int level, int k) {
  derp_ctx* c = q->c;
  const int W = c->LW[level], H = c->LH[level];
  const size_t n = (size_t)W * H;
  const int radius = temporal_space_radius(q, level);
  const int t = q->owned[k];
  int lo, hi;
  seq_window(t, q->first, q->last, q->opt.time_radius, &lo, &hi);
  std::vector<const void*> wg(hi - lo + 1);
  std::vector<const float*> wi(hi - lo + 1);
  std::vector<const uint8_t*> wmk(hi - lo + 1);
  for (int u = lo; u <= hi; ++u) {
    void *pc, *pd, *pm;
    size_t b;
    TRY(seq_buffer(q, u, level, 0, &pc, &b));
    TRY(seq_buffer(q, u, level, 2, &pd, &b));
    if (q->opt.use_foreground_masks) {  // mask = fg & fov of each frame (TemporalBilateralFilter.cpp:150-160)
      TRY(seq_buffer(q, u, level, 1, &pm, &b));
      uint8_t* wm = q->winMask.as<uint8_t>() + (size_t)(u - lo) * n * c->D;
      hipLaunchKernelGGL(k_and_masks, dim3(flat_grid(n), c->D), dim3(256), 0, c->stream, q->fov.as<uint8_t>(),
                         (const uint8_t*)pm, c->dst2src.as<int>(), 0, n, wm);
      KCHECK(c);
      pm = wm;
    } else {
      pm = q->fov.p;  // generateAllPassMasks & fov
    }
    wg[u - lo] = pc;
    wi[u - lo] = reinterpret_cast<const float*>(pd);
    wmk[u - lo] = reinterpret_cast<const uint8_t*>(pm);
  }
  // weights (b, g, b): the reference passes FLAGS_weight_b for the third channel (TemporalBilateralFilter.cpp:176-178)
  TRY(temporal_launch(c, wg.data(), wi.data(), wmk.data(), hi - lo + 1, t - lo, W, H, c->D, q->opt.sigma, radius,
                      q->opt.weight_b, q->opt.weight_g, q->opt.weight_b, q->filtered[k].as<float>(), c->dst2src.as<int>()));
  q->filteredAt[k] = level;
  HIPCHK(c, hipEventRecord(q->filteredEv[k], c->stream));
  return 0;
}
}  // namespace

// One owned frame's filter AHEAD of derp_seq_level_filter — as soon as every frame of its window holds its raw
// level (owned frames: computed at this level; halo frames: exchanged), which for the frames in the middle of a
// rank's chunk is long before the last frame of the level is computed. The result waits in the frame's scratch
// (derp_seq_download_filtered) — nothing is overwritten: the Transfer still happens in derp_seq_level_filter, after
// every frame is filtered. Returns 0 = filtered, 2 = not yet possible (no error recorded), 1 = error.
int derp_seq_level_filter_frame(derp_seq* q, int level, int frame) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  HIPCHK(c, hipSetDevice(c->device));
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0) {
    return fail(c, "frame %d is not owned by rank %d", frame, q->rank);
  }
  if (!q->opt.do_temporal_filter || q->streaming) {
    return 2;
  }
  if (q->filteredAt[k] == level) {
    return 0;
  }
  int lo, hi;
  seq_window(frame, q->first, q->last, q->opt.time_radius, &lo, &hi);
  bool needsHalo = false;
  for (int u = lo; u <= hi; ++u) {
    const int ku = owned_index(q, u);
    if (ku >= 0 ? q->computedAt[ku] != level : q->levelExchanged != level) {
      return 2;
    }
    needsHalo |= ku < 0;
  }
  if (needsHalo) {
    TRY(seq_join_exchange(q));
  }
  Span sp(c, ST_TEMPORAL, level);
  TRY(seq_fov_masks(q, level));
  return seq_filter_frame(q, level, k);
}

// A destination's filtered level of an owned frame straight from the filter's scratch, on the library's copy
// stream behind the frame's filter kernel: the compute stream keeps running the frames that follow.
int derp_seq_download_filtered(derp_seq* q, int frame, int level, int d, float* out) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  TRY(check_level(c, level));
  const int k = owned_index(q, frame);
  if (k < 0 || d < -1 || d >= c->D || !out) {
    return fail(c, "bad frame / destination index / null output");
  }
  if (q->streaming || q->filteredAt[k] != level) {
    return fail(c, "frame %d has no filtered level %d in its scratch (derp_seq_level_filter_frame)", frame, level);
  }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = npx(c, level);
  HIPCHK(c, hipStreamWaitEvent(c->copyStream, q->filteredEv[k], 0));
  // dst = -1: every destination's plane in one copy, [D][h*w]
  HIPCHK(c, hipMemcpyAsync(out, q->filtered[k].as<float>() + (size_t)std::max(d, 0) * n, (d < 0 ? c->D : 1) * n * sizeof(float),
                           hipMemcpyDeviceToHost, c->copyStream));
  HIPCHK(c, hipStreamSynchronize(c->copyStream));
  return 0;
}

int derp_seq_level_filter(derp_seq* q, int level) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  HIPCHK(c, hipSetDevice(c->device));
  TRY(check_level(c, level));
  if (!q->opt.do_temporal_filter) {
    return 0;
  }
  // the filter reads the raw level of every window frame: a skipped or reordered phase would filter stale data
  if (q->levelReady != level) {
    return fail(c, "derp_seq_level_filter(%d): derp_seq_level_compute(%d) has not completed for every owned frame", level, level);
  }
  if (!q->halo.empty() && q->levelExchanged != level) {
    return fail(c, "derp_seq_level_filter(%d): the halo frames' level has not been exchanged (derp_seq_level_exchange, or "
                   "derp_seq_mark_exchanged after an external transport moved it)", level);
  }
  const int W = c->LW[level], H = c->LH[level];
  const size_t n = (size_t)W * H;
  Span sp(c, ST_TEMPORAL, level);  // masks + temporal kernels + write-back of every owned frame
  TRY(seq_fov_masks(q, level));
  if (q->streaming) {
    TRY(seq_join_exchange(q));
    const int radius = temporal_space_radius(q, level);
    TRY(stream_filter_level(q, level, W, H, radius));
    q->levelReady = -1;
    return 0;
  }
  // frames whose windows hold only owned frames first: they run beside the exchange (xStream); then the compute stream
  // waits for the halo frames' levels and filters the rest
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      TRY(seq_join_exchange(q));
    }
    for (int k = 0; k < (int)q->owned.size(); ++k) {
      if (q->filteredAt[k] == level) {  // filtered ahead of time (derp_seq_level_filter_frame) or in pass 0
        continue;
      }
      int lo, hi;
      seq_window(q->owned[k], q->first, q->last, q->opt.time_radius, &lo, &hi);
      bool allOwned = true;
      for (int u = lo; allOwned && u <= hi; ++u) {
        allOwned = owned_index(q, u) >= 0;
      }
      if (pass == 1 || allOwned) {
        TRY(seq_filter_frame(q, level, k));
      }
    }
  }
  // "Transfer" (pipeline.py:397-408): every owned frame is filtered before any raw level is overwritten — and (the join
  // above) no send of the exchange still reads one
  for (int k = 0; k < (int)q->owned.size(); ++k) {
    SlotView v = slot_view(c, k);
    HIPCHK(c, hipMemcpyAsync((*v.disp)[level].p, q->filtered[k].p, n * c->D * sizeof(float), hipMemcpyDeviceToDevice,
                             c->stream));
    q->computedAt[k] = -1;  // the slot holds the filtered level now: no window may read it as a raw level
  }
  q->levelReady = -1;
  return 0;
}

int derp_seq_run(derp_seq* q, int level_start, int level_end_) {
  if (!q) {
    return 1;
  }
  derp_ctx* c = q->c;
  if (level_start < level_end_) {
    return fail(c, "Check failed: level_start >= level_end (%d vs %d)", level_start, level_end_);
  }
  if (!q->plan.empty() && q->transport != SEQ_RCCL) {
    return fail(c, "derp_seq_run drives all three phases itself and needs the RCCL transport; with loopback / "
                   "external transports call derp_seq_level_compute / _exchange / _filter per level");
  }
  for (int level = level_start; level >= level_end_; --level) {
    TRY(derp_seq_level_compute(q, level));
    TRY(derp_seq_level_exchange(q, level));
    TRY(derp_seq_level_filter(q, level));
  }
  return 0;
}

int derp_seq_stats(derp_seq* q, uint64_t* bytes_sent, uint64_t* bytes_received, double* exchange_ms) {
  if (!q) {
    return 1;
  }
  HIPCHK(q->c, hipStreamSynchronize(q->c->stream));
  seq_drain_spans(q);
  if (bytes_sent) {
    *bytes_sent = q->bytesSent;
  }
  if (bytes_received) {
    *bytes_received = q->bytesRecv;
  }
  if (exchange_ms) {
    *exchange_ms = q->exchangeMs;
  }
  return 0;
}

// the part of derp_seq_stats' exchange_ms the compute stream stood still for (the rest ran beside the filter of the
// frames whose windows are local)
int derp_seq_exchange_exposed_ms(derp_seq* q, double* exposed_ms) {
  if (!q) {
    return 1;
  }
  HIPCHK(q->c, hipStreamSynchronize(q->c->stream));
  seq_drain_spans(q);
  if (exposed_ms) {
    *exposed_ms = q->exchangeExposedMs;
  }
  return 0;
}

int derp_seq_stats_reset(derp_seq* q) {
  if (!q) {
    return 1;
  }
  HIPCHK(q->c, hipStreamSynchronize(q->c->stream));
  seq_drain_spans(q);
  q->bytesSent = q->bytesRecv = 0;
  q->exchangeMs = 0;
  q->exchangeExposedMs = 0;
  return 0;
}

}  // extern "C"
