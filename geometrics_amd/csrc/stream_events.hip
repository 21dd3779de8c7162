// Host-side stream/event utilities of the data-parallel step (no kernels).
//
// The step is replayed as ONE HIP graph, and the gradient all-reduce has to start as soon as the end-of-pass reduction
// launch -- a node in the MIDDLE of that graph -- has written the bucket, while the rest of the graph (the first layer's
// input-gradient product) keeps running.  HIP expresses that with an EXTERNAL event: hipEventRecordWithFlags(...,
// hipEventRecordExternal) on a capturing stream becomes an event-record NODE of the graph (instead of an intra-capture
// dependency), and a stream outside the graph waits for it with hipStreamWaitEvent.  PyTorch-ROCm refuses
// torch.cuda.Event(external=True) ("External events are disallowed in rocm"), hence these four entry points on raw
// hipEvent_t / hipStream_t handles.  Outside a capture geom_event_record is a plain hipEventRecord.
#include "geom_common.h"

extern "C" int geom_event_create(void **event)
{
    if (!event) return GEOM_EINVAL;
    hipEvent_t e = nullptr;
    const hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    *event = e;
    return (int)err;
}

extern "C" int geom_event_destroy(void *event)
{
    return event ? (int)hipEventDestroy(static_cast<hipEvent_t>(event)) : 0;
}

extern "C" int geom_event_record(void *event, void *stream)
{
    if (!event) return GEOM_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    hipError_t err = hipStreamIsCapturing(s, &status);
    if (err != hipSuccess) return (int)err;
    const unsigned flags = status == hipStreamCaptureStatusActive ? hipEventRecordExternal : 0u;   // the flag is invalid outside a capture
    return (int)hipEventRecordWithFlags(static_cast<hipEvent_t>(event), s, flags);
}

extern "C" int geom_stream_wait_event(void *stream, void *event)
{
    if (!event) return GEOM_EINVAL;
    return (int)hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0);
}
