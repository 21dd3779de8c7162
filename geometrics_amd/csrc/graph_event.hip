// An event-record node in the MIDDLE of a captured step (host-side graph surgery; one empty marker kernel).
//
// The data-parallel step is replayed as ONE HIP graph, and its gradient all-reduce -- issued from another stream, outside
// the graph -- has to start as soon as the end-of-pass reduction launch has written the bucket, while the rest of the
// graph (the first layer's input-gradient product) keeps running.  A graph expresses that with an EVENT-RECORD NODE.
// Stream capture cannot create one on this stack (torch refuses external events on ROCm, and
// hipEventRecordWithFlags(hipEventRecordExternal) returns hipErrorInvalidValue during a capture with the HIP runtime it
// bundles: tools/probe/external_event.py), and cutting the step into two graphs costs ~15 us of launch latency per cut
// (measured: profiles/r04_dp_fixed_cost.txt).  So the capture drops a MARKER -- an empty kernel, geom_graph_marker -- where
// the event belongs, and geom_graph_event_at_marker replaces that node by hipGraphAddEventRecordNode before the graph is
// instantiated: the marker's predecessors become the record node's, its successors are re-attached to its predecessors.
#include <vector>

#include "geom_common.h"

namespace {

__global__ void graph_marker_kernel() {}

} // namespace

extern "C" int geom_graph_marker(void *stream)
{
    hipLaunchKernelGGL(graph_marker_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream));
    return geom::launch_status();
}

extern "C" int geom_graph_event_at_marker(void *graph, void *event, int *replaced)
{
    if (!graph || !event || !replaced) return GEOM_EINVAL;
    *replaced = 0;
    hipGraph_t g = static_cast<hipGraph_t>(graph);
    size_t n = 0;
    hipError_t err = hipGraphGetNodes(g, nullptr, &n);
    if (err != hipSuccess) return (int)err;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && (err = hipGraphGetNodes(g, nodes.data(), &n)) != hipSuccess) return (int)err;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if ((err = hipGraphNodeGetType(nodes[i], &type)) != hipSuccess) return (int)err;
        if (type != hipGraphNodeTypeKernel) continue;
        hipKernelNodeParams params;
        if ((err = hipGraphKernelNodeGetParams(nodes[i], &params)) != hipSuccess) return (int)err;
        if (params.func != reinterpret_cast<void *>(graph_marker_kernel)) continue;
        size_t nd = 0, ns = 0;
        if ((err = hipGraphNodeGetDependencies(nodes[i], nullptr, &nd)) != hipSuccess) return (int)err;
        std::vector<hipGraphNode_t> deps(nd);
        if (nd && (err = hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd)) != hipSuccess) return (int)err;
        if ((err = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &ns)) != hipSuccess) return (int)err;
        std::vector<hipGraphNode_t> succ(ns);
        if (ns && (err = hipGraphNodeGetDependentNodes(nodes[i], succ.data(), &ns)) != hipSuccess) return (int)err;
        hipGraphNode_t record;
        if ((err = hipGraphAddEventRecordNode(&record, g, deps.data(), nd, static_cast<hipEvent_t>(event))) != hipSuccess)
            return (int)err;
        for (size_t s = 0; s < ns; ++s)
            for (size_t d = 0; d < nd; ++d)
                if ((err = hipGraphAddDependencies(g, &deps[d], &succ[s], 1)) != hipSuccess) return (int)err;
        if ((err = hipGraphDestroyNode(nodes[i])) != hipSuccess) return (int)err;
        ++*replaced;
    }
    return 0;
}
