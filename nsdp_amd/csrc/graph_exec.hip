// Multi-stream replay of a stream-captured train step: the host side of a step taken off the critical path.
//
// A TDNet train step is ~1000 kernel launches enqueued one Python / ctypes call at a time (17-24 ms of host time per
// step, whatever the batch).  hipGraphLaunch of the captured step removes the host cost but loses the schedule: on this
// ROCm the replay runs the weight-gradient branch (hip_linear's side stream) and the geometry branch serialised with the
// main chain -- 47.7 ms against 44.4 ms eager at B = 32, for every DEBUG_HIP_*GRAPH* knob (profiles/r3_graph_probe.txt).
//
// This executor takes the SAME captured hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph(): our kernels
// and ATen's, memsets, copies -- whatever the step enqueued) and replays it itself: the nodes are put into a topological
// order once, every node is assigned to one of a few real HIP streams by chain decomposition (a node continues the stream
// whose last node it depends on; a fork opens / re-uses a side stream), cross-stream edges become hipEventRecord /
// hipStreamWaitEvent pairs (redundant waits are dropped: streams are in order), and a replay is one C loop of
// hipLaunchKernel / hipMemsetAsync / hipMemcpy3DAsync calls -- ~4 us per node, no Python, the eager schedule's
// concurrency.  The node parameters (kernel argument blocks included) stay owned by the graph, which the caller keeps
// alive; memory is the capture's private pool, exactly as with hipGraphLaunch.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

struct ExecNode {
  hipGraphNodeType type;
  hipKernelNodeParams kp;
  hipMemsetParams ms;
  hipMemcpy3DParms cp;
  hipGraphExec_t sub = nullptr;      // nodes replayed through a one-node graph of their own (see one_node_exec)
  hipGraph_t sub_graph = nullptr;
  int stream = 0;                    // index into GraphExec::streams (0 = the caller's stream)
  int record = -1;                   // event to record after this node, or -1
  std::vector<int> waits;     // events this node's stream waits for before the node
  hipEvent_t t0 = nullptr, t1 = nullptr;      // launch_timed: timing events around this kernel (created on first use)
};

struct GraphExec {
  hipGraph_t graph;
  std::vector<ExecNode> nodes;
  std::vector<hipStream_t> side;      // owned side streams (stream index s >= 1 -> side[s - 1])
  std::vector<hipEvent_t> events;
  hipEvent_t begin;
  std::vector<hipEvent_t> tail_events;
  int n_kernels = 0, n_cross = 0, n_streams = 1, n_sub = 0;
};

// A node this executor cannot re-issue from its parameters (1-D memcpy nodes have no public getter on this ROCm; event /
// host nodes) is replayed through a graph of its own: a clone of the captured graph with every other node removed,
// launched with hipGraphLaunch on the node's stream.  A handful of nodes per step; built once.
hipError_t one_node_exec(hipGraph_t graph, const std::vector<hipGraphNode_t> &handles, hipGraphNode_t keep, hipGraph_t *out_graph,
                         hipGraphExec_t *out_exec) {
  hipGraph_t clone = nullptr;
  hipError_t e = hipGraphClone(&clone, graph);
  if (e != hipSuccess) return e;
  for (hipGraphNode_t h : handles) {
    if (h == keep) continue;
    hipGraphNode_t c = nullptr;
    e = hipGraphNodeFindInClone(&c, h, clone);
    if (e == hipSuccess) e = hipGraphDestroyNode(c);
    if (e != hipSuccess) {
      (void)hipGraphDestroy(clone);
      return e;
    }
  }
  e = hipGraphInstantiate(out_exec, clone, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(clone);
    return e;
  }
  *out_graph = clone;
  return hipSuccess;
}

// `what` == nullptr: the caller has already written the specific message (kept as it is)
int fail(GraphExec *g, const char *what, hipError_t e) {
  if (what) nsdp::set_error("graph_exec: %s: %s", what, hipGetErrorString(e));
  if (g) {
    for (auto &x : g->nodes) {
      if (x.sub) (void)hipGraphExecDestroy(x.sub);
      if (x.sub_graph) (void)hipGraphDestroy(x.sub_graph);
      if (x.t0) (void)hipEventDestroy(x.t0);
      if (x.t1) (void)hipEventDestroy(x.t1);
    }
    for (auto s : g->side) (void)hipStreamDestroy(s);
    for (auto ev : g->events) (void)hipEventDestroy(ev);
    for (auto ev : g->tail_events) (void)hipEventDestroy(ev);
    if (g->begin) (void)hipEventDestroy(g->begin);
    delete g;
  }
  return static_cast<int>(e == hipSuccess ? hipErrorUnknown : e);
}

}  // namespace

extern "C" {

int nsdp_graph_exec_create(void *graph_v, int max_streams, void **out) {
  NSDP_REQUIRE(graph_v && out, "graph_exec_create: null argument");
  NSDP_REQUIRE(max_streams >= 1 && max_streams <= 8, "graph_exec_create: max_streams=%d must be in [1, 8]", max_streams);
  hipGraph_t graph = static_cast<hipGraph_t>(graph_v);
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return fail(nullptr, "hipGraphGetNodes", e);
  NSDP_REQUIRE(n > 0, "graph_exec_create: the graph is empty");
  std::vector<hipGraphNode_t> handles(n);
  e = hipGraphGetNodes(graph, handles.data(), &n);
  if (e != hipSuccess) return fail(nullptr, "hipGraphGetNodes", e);
  // creation order of the handles = capture order; index nodes by it
  std::vector<std::pair<hipGraphNode_t, int>> by_handle(n);
  for (size_t i = 0; i < n; ++i) by_handle[i] = {handles[i], static_cast<int>(i)};
  std::sort(by_handle.begin(), by_handle.end());
  auto index_of = [&](hipGraphNode_t h) -> int {
    auto it = std::lower_bound(by_handle.begin(), by_handle.end(), std::make_pair(h, -1));
    return (it != by_handle.end() && it->first == h) ? it->second : -1;
  };
  std::vector<std::vector<int>> deps(n), users(n);
  for (size_t i = 0; i < n; ++i) {
    size_t nd = 0;
    e = hipGraphNodeGetDependencies(handles[i], nullptr, &nd);
    if (e != hipSuccess) return fail(nullptr, "hipGraphNodeGetDependencies", e);
    if (!nd) continue;
    std::vector<hipGraphNode_t> d(nd);
    e = hipGraphNodeGetDependencies(handles[i], d.data(), &nd);
    if (e != hipSuccess) return fail(nullptr, "hipGraphNodeGetDependencies", e);
    for (size_t j = 0; j < nd; ++j) {
      const int k = index_of(d[j]);
      NSDP_REQUIRE(k >= 0, "graph_exec_create: a dependency is not a node of the graph");
      deps[i].push_back(k);
      users[k].push_back(static_cast<int>(i));
    }
  }
  // topological order, ties broken by capture order (Kahn with a min-heap on the node index)
  std::vector<int> indeg(n), order;
  order.reserve(n);
  std::vector<int> heap;
  for (size_t i = 0; i < n; ++i) {
    indeg[i] = static_cast<int>(deps[i].size());
    if (!indeg[i]) heap.push_back(static_cast<int>(i));
  }
  auto cmp = [](int a, int b) { return a > b; };
  std::make_heap(heap.begin(), heap.end(), cmp);
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    const int u = heap.back();
    heap.pop_back();
    order.push_back(u);
    for (int v : users[u])
      if (--indeg[v] == 0) {
        heap.push_back(v);
        std::push_heap(heap.begin(), heap.end(), cmp);
      }
  }
  NSDP_REQUIRE(order.size() == n, "graph_exec_create: the graph has a cycle");

  GraphExec *g = new GraphExec();
  g->graph = graph;
  g->begin = nullptr;
  g->nodes.resize(n);
  std::vector<int> pos(n);                 // position of graph node i in the replay order
  for (size_t p = 0; p < n; ++p) pos[order[p]] = static_cast<int>(p);
  // ---- node parameters ------------------------------------------------------------------------------------------
  for (size_t p = 0; p < n; ++p) {
    ExecNode &x = g->nodes[p];
    hipGraphNode_t h = handles[order[p]];
    x.record = -1;
    x.stream = 0;
    x.sub = nullptr;
    x.sub_graph = nullptr;
    memset(&x.kp, 0, sizeof(x.kp));
    memset(&x.cp, 0, sizeof(x.cp));
    bool own_graph = false;
    e = hipGraphNodeGetType(h, &x.type);
    if (e != hipSuccess) return fail(g, "hipGraphNodeGetType", e);
    switch (x.type) {
      case hipGraphNodeTypeKernel:
        e = hipGraphKernelNodeGetParams(h, &x.kp);
        if (e != hipSuccess) return fail(g, "hipGraphKernelNodeGetParams", e);
        if (!x.kp.func || (!x.kp.kernelParams && !x.kp.extra)) {
          nsdp::set_error("graph_exec_create: kernel node %zu has no function / argument block", p);
          return fail(g, nullptr, hipErrorInvalidValue);
        }
        if (!x.kp.kernelParams) {      // (module launches with an `extra` buffer: not produced by this step)
          nsdp::set_error("graph_exec_create: kernel node %zu passes its arguments through `extra` (unsupported)", p);
          return fail(g, nullptr, hipErrorNotSupported);
        }
        ++g->n_kernels;
        break;
      case hipGraphNodeTypeMemset:
        e = hipGraphMemsetNodeGetParams(h, &x.ms);
        if (e != hipSuccess) return fail(g, "hipGraphMemsetNodeGetParams", e);
        // hipMemset2DAsync has byte semantics only: a 2-D memset of 2- / 4-byte elements (whose pattern bytes may differ)
        // is replayed as the node it is, through a graph of its own
        if (x.ms.height > 1 && x.ms.elementSize > 1) own_graph = true;
        break;
      case hipGraphNodeTypeMemcpy:
        // (a 1-D memcpy node answers with an empty 3-D description or an error: replay it through its own graph)
        // (stream capture records copies as 1-D memcpy nodes, which have no public parameter getter on this ROCm -- the
        // 3-D getter answers with a description hipMemcpy3DAsync rejects: every copy node is replayed through a graph of
        // its own; there are a handful per step)
        own_graph = true;
        break;
      case hipGraphNodeTypeEmpty:
        break;
      default:
        own_graph = true;
        break;
    }
    if (own_graph) {
      e = one_node_exec(graph, handles, h, &x.sub_graph, &x.sub);
      if (e != hipSuccess) {
        nsdp::set_error("graph_exec_create: node %zu (type %d) could not be isolated into a graph of its own: %s", p,
                        static_cast<int>(x.type), hipGetErrorString(e));
        return fail(g, nullptr, e);
      }
      ++g->n_sub;
    }
  }
  // ---- streams: chain decomposition -----------------------------------------------------------------------------
  std::vector<int> tail(max_streams, -1);          // replay position of the last node on each stream
  std::vector<std::vector<int>> dpos(n);           // dependencies as replay positions
  for (size_t p = 0; p < n; ++p)
    for (int d : deps[order[p]]) dpos[p].push_back(pos[d]);
  // Which successor continues a node's stream: the one with the LONGEST chain behind it (its "heir"), not the one that happened
  // to be captured first -- a short branch captured ahead of the step (the next batch's geometry of a pipelined step) otherwise
  // takes the main chain's stream and hangs the step and its weight gradients on one side stream (44.0 against 40.3 ms at B = 32).
  // NSDP_GRAPH_HEIR=0: first come, first served (rounds 3-4).
  static const bool by_height = !(getenv("NSDP_GRAPH_HEIR") && atoi(getenv("NSDP_GRAPH_HEIR")) == 0);
  std::vector<int> height(n, 1), heir(n, -1);
  for (size_t q = n; q-- > 0;)
    for (int d : dpos[q]) {      // q is a successor of d; positions are topological, so height[q] is final here
      if (height[q] + 1 > height[d]) height[d] = height[q] + 1;
    }
  for (size_t q = 0; q < n; ++q)
    for (int d : dpos[q])
      if (heir[d] < 0 || height[q] > height[heir[d]]) heir[d] = static_cast<int>(q);
  int used = 1, rr = 0;
  for (size_t p = 0; p < n; ++p) {
    ExecNode &x = g->nodes[p];
    int s = -1;
    if (dpos[p].empty()) {
      s = tail[0] < 0 ? 0 : -1;                    // the first root is the main chain; later roots are forks
    } else {
      for (int c = 0; c < used && s < 0; ++c)
        for (int d : dpos[p])
          if (tail[c] == d && (!by_height || heir[d] == static_cast<int>(p))) { s = c; break; }
    }
    if (s < 0) {                                   // a fork: a fresh side stream, else round-robin over the side streams
      if (used < max_streams) s = used++;
      else if (max_streams > 1) s = 1 + (rr++ % (max_streams - 1));
      else s = 0;
    }
    x.stream = s;
    tail[s] = static_cast<int>(p);
  }
  g->n_streams = used;
  // ---- cross-stream edges -> events (a wait on event E of stream S also covers everything S ran before E) ----------
  std::vector<std::vector<int>> covered(used, std::vector<int>(used, -1));     // [dst][src]: latest src position waited for
  for (size_t p = 0; p < n; ++p) {
    ExecNode &x = g->nodes[p];
    std::vector<int> ds = dpos[p];
    std::sort(ds.begin(), ds.end(), [](int a, int b) { return a > b; });       // latest first: covers the earlier ones
    for (int d : ds) {
      const int src = g->nodes[d].stream;
      if (src == x.stream || covered[x.stream][src] >= d) continue;
      if (g->nodes[d].record < 0) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return fail(g, "hipEventCreateWithFlags", e);
        g->nodes[d].record = static_cast<int>(g->events.size());
        g->events.push_back(ev);
      }
      x.waits.push_back(g->nodes[d].record);
      covered[x.stream][src] = d;
      ++g->n_cross;
    }
  }
  // (branch streams at normal priority: hipStreamCreateWithPriority at EITHER end of the device's range made the B = 32 step
  // 64 ms instead of 43 ms -- docs/EXPERIMENTS.md)
  for (int s = 1; s < used; ++s) {
    hipStream_t st = nullptr;
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(g, "hipStreamCreateWithFlags", e);
    g->side.push_back(st);
    hipEvent_t ev = nullptr;
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return fail(g, "hipEventCreateWithFlags", e);
    g->tail_events.push_back(ev);
  }
  e = hipEventCreateWithFlags(&g->begin, hipEventDisableTiming);
  if (e != hipSuccess) return fail(g, "hipEventCreateWithFlags", e);
  *out = g;
  return 0;
}

int nsdp_graph_exec_info(void *handle, int *nodes, int *kernels, int *streams, int *cross_edges, int *own_graph_nodes) {
  NSDP_REQUIRE(handle, "graph_exec_info: null handle");
  const GraphExec *g = static_cast<const GraphExec *>(handle);
  if (nodes) *nodes = static_cast<int>(g->nodes.size());
  if (kernels) *kernels = g->n_kernels;
  if (streams) *streams = g->n_streams;
  if (cross_edges) *cross_edges = g->n_cross;
  if (own_graph_nodes) *own_graph_nodes = g->n_sub;
  return 0;
}

static int launch_impl(GraphExec *g, hipStream_t main, const char *timed_substr);

int nsdp_graph_exec_launch(void *handle, void *stream) {
  NSDP_REQUIRE(handle, "graph_exec_launch: null handle");
  return launch_impl(static_cast<GraphExec *>(handle), nsdp::as_stream(stream), nullptr);
}

int nsdp_graph_exec_launch_timed(void *handle, void *stream, const char *name_substr, long long *launches, double *total_ms) {
  NSDP_REQUIRE(handle && name_substr && launches && total_ms, "graph_exec_launch_timed: null argument");
  GraphExec *g = static_cast<GraphExec *>(handle);
  hipStream_t main = nsdp::as_stream(stream);
  const int rc = launch_impl(g, main, name_substr);
  if (rc) return rc;
  NSDP_HIP_TRY(hipStreamSynchronize(main));
  long long n = 0;
  double ms = 0.0;
  for (ExecNode &x : g->nodes) {
    if (!x.t0) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, x.t0, x.t1) == hipSuccess) {
      ++n;
      ms += t;
    }
    (void)hipEventDestroy(x.t0);
    (void)hipEventDestroy(x.t1);
    x.t0 = x.t1 = nullptr;
  }
  (void)hipGetLastError();
  *launches = n;
  *total_ms = ms;
  return 0;
}

static int launch_impl(GraphExec *g, hipStream_t main, const char *timed_substr) {
  auto st_of = [&](int s) { return s == 0 ? main : g->side[s - 1]; };
  if (g->n_streams > 1) {      // side streams start behind everything the caller's stream has been given so far
    NSDP_HIP_TRY(hipEventRecord(g->begin, main));
    for (int s = 1; s < g->n_streams; ++s) NSDP_HIP_TRY(hipStreamWaitEvent(st_of(s), g->begin, 0));
  }
  for (ExecNode &x : g->nodes) {
    hipStream_t st = st_of(x.stream);
    for (int w : x.waits) NSDP_HIP_TRY(hipStreamWaitEvent(st, g->events[w], 0));
    if (x.sub) {
      NSDP_HIP_TRY(hipGraphLaunch(x.sub, st));
    } else switch (x.type) {
      case hipGraphNodeTypeKernel: {
        bool timed = false;
        if (timed_substr) {
          const char *nm = hipKernelNameRefByPtr(x.kp.func, st);
          timed = nm && strstr(nm, timed_substr);
          if (!nm) (void)hipGetLastError();
        }
        if (timed) {
          NSDP_HIP_TRY(hipEventCreate(&x.t0));
          NSDP_HIP_TRY(hipEventCreate(&x.t1));
          NSDP_HIP_TRY(hipEventRecord(x.t0, st));
        }
        NSDP_HIP_TRY(hipLaunchKernel(x.kp.func, x.kp.gridDim, x.kp.blockDim, x.kp.kernelParams, x.kp.sharedMemBytes, st));
        if (timed) NSDP_HIP_TRY(hipEventRecord(x.t1, st));
        break;
      }
      case hipGraphNodeTypeMemset:
        if (x.ms.height <= 1) {
          if (x.ms.elementSize == 4) NSDP_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(x.ms.dst), static_cast<int>(x.ms.value), x.ms.width, st));
          else if (x.ms.elementSize == 2) NSDP_HIP_TRY(hipMemsetD16Async(reinterpret_cast<hipDeviceptr_t>(x.ms.dst), static_cast<unsigned short>(x.ms.value), x.ms.width, st));
          else NSDP_HIP_TRY(hipMemsetAsync(x.ms.dst, static_cast<int>(x.ms.value), x.ms.width * (x.ms.elementSize ? x.ms.elementSize : 1), st));
        } else {
          NSDP_HIP_TRY(hipMemset2DAsync(x.ms.dst, x.ms.pitch, static_cast<int>(x.ms.value), x.ms.width * (x.ms.elementSize ? x.ms.elementSize : 1), x.ms.height, st));
        }
        break;
      case hipGraphNodeTypeMemcpy:
        NSDP_HIP_TRY(hipMemcpy3DAsync(&x.cp, st));
        break;
      default:
        break;
    }
    if (x.record >= 0) NSDP_HIP_TRY(hipEventRecord(g->events[x.record], st));
  }
  for (int s = 1; s < g->n_streams; ++s) {      // join: the caller's stream continues behind every branch
    NSDP_HIP_TRY(hipEventRecord(g->tail_events[s - 1], st_of(s)));
    NSDP_HIP_TRY(hipStreamWaitEvent(main, g->tail_events[s - 1], 0));
  }
  return 0;
}

int nsdp_graph_exec_destroy(void *handle) {
  if (!handle) return 0;
  GraphExec *g = static_cast<GraphExec *>(handle);
  for (auto s : g->side) {
    (void)hipStreamSynchronize(s);
    (void)hipStreamDestroy(s);
  }
  for (auto &x : g->nodes) {
    if (x.sub) (void)hipGraphExecDestroy(x.sub);
    if (x.sub_graph) (void)hipGraphDestroy(x.sub_graph);
    if (x.t0) (void)hipEventDestroy(x.t0);
    if (x.t1) (void)hipEventDestroy(x.t1);
  }
  for (auto ev : g->events) (void)hipEventDestroy(ev);
  for (auto ev : g->tail_events) (void)hipEventDestroy(ev);
  if (g->begin) (void)hipEventDestroy(g->begin);
  delete g;
  return 0;
}

}  // extern "C"
