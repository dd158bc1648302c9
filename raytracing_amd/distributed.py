"""Sharding one frame over ranks (one process per GPU) and putting it back together.

The reference has a single OpenMP loop over image rows (src/Scene.cpp:214-215).  Here rank r of G
owns rows r, r+G, r+2G, ... (row-interleaved, so the cheap sky rows and the expensive bunny rows
are spread evenly), renders them compactly into its own device buffer, and the buffers meet on
rank 0 in one exchange: a gather of equal-sized blocks over RCCL (point-to-point xGMI transfers into
rank 0), followed by a de-interleave kernel.  The scene is replicated; nothing else is exchanged.

torch is used for device buffers, streams and torch.distributed only.
"""
import ctypes as C

from .rtx import Frame


def rows_per_rank(height, world_size):
    """Rows in every rank's buffer: ceil(height / world_size).  Ranks whose share is one row short
    (height not divisible) leave their last buffer row unused."""
    return (height + world_size - 1) // world_size


def rank_frame(width, height, depth, rank, world_size):
    """The rtx_frame describing rank's share of a width x height frame."""
    return Frame(width, height, depth, rank, world_size)


def owned_rows(height, rank, world_size):
    return list(range(rank, height, world_size))


class ShardedRenderer:
    """Renders row-interleaved shares of a frame and gathers them on rank 0.

    ctx: a committed raytracing_amd Context on this rank's GPU.  `process_group` None = single GPU
    (no communication at all).  Buffers are allocated once; render() enqueues on the renderer's own
    streams and returns the assembled frame tensor on rank 0 (None elsewhere).
    """

    def __init__(self, ctx, width, height, depth, rank=0, world_size=1, device=None, n_buffers=2):
        import torch
        self.torch = torch
        self.ctx, self.width, self.height, self.depth = ctx, width, height, depth
        self.rank, self.world = rank, world_size
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.rpr = rows_per_rank(height, world_size)
        self.frame_desc = rank_frame(width, height, depth, rank, world_size)
        row_bytes = width * 3
        self.local = [torch.zeros(self.rpr * row_bytes, dtype=torch.uint8, device=self.device) for _ in range(n_buffers)]
        self.gathered = None
        self.frames = None
        if rank == 0 and world_size > 1:
            self.gathered = [torch.zeros(world_size * self.rpr * row_bytes, dtype=torch.uint8, device=self.device)
                             for _ in range(n_buffers)]
            self.frames = [torch.zeros(height * row_bytes, dtype=torch.uint8, device=self.device) for _ in range(n_buffers)]
        self.pending = [None] * n_buffers   # ranks > 0: the gather that still reads local[k]
        self.released = [None] * n_buffers  # rank 0: event after which local[k] / gathered[k] / frames[k] may be reused
        self.step = 0
        # Streams of our own: rtx_render_device treats a NULL stream as "the context's stream", and
        # torch's default stream IS the NULL stream.  Renders go to `stream`; the collective and the
        # de-interleave go to `comm`, which waits for the render it ships, so that the exchange of
        # frame k runs while frame k+1 is being rendered (buffers are double).
        self.stream = torch.cuda.Stream(self.device)
        self.comm = torch.cuda.Stream(self.device)
        # the zero-fills above were enqueued on the caller's current stream: nothing of ours may overtake them
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self.comm.wait_stream(torch.cuda.current_stream(self.device))

    def render(self, cam):
        """One frame.  Enqueues the render of this rank's rows, the gather and (rank 0) the
        de-interleave.  Returns (frame_tensor_or_None, work_handle_or_None); the frame is complete once
        finish() has been called (or the `comm` stream has been waited for)."""
        torch = self.torch
        k = self.step % len(self.local)
        self.step += 1
        with torch.cuda.stream(self.stream):
            # the buffer's previous exchange must have drained before it is rendered into again
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None
            if self.released[k] is not None:
                self.stream.wait_event(self.released[k])
                self.released[k] = None
            self.ctx.render_device(cam, self.frame_desc, self.local[k].data_ptr(), self.stream.cuda_stream)
            if self.world == 1:
                return self.local[k], None
            rendered = torch.cuda.Event()
            rendered.record(self.stream)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(rendered)  # NCCL work is ordered after the current stream at the time of the call
            if self.rank == 0:
                chunks = list(self.gathered[k].chunk(self.world))
                work = self._gather(self.local[k], chunks)
                work.wait()  # orders `comm` after the gather; does not block the host
                self.ctx.deinterleave_device(self.gathered[k].data_ptr(), self.world, self.rpr, self.width, self.height,
                                             self.frames[k].data_ptr(), self.comm.cuda_stream)
                self.released[k] = torch.cuda.Event()
                self.released[k].record(self.comm)
                return self.frames[k], None
            work = self._gather(self.local[k], None)
            self.pending[k] = work
            return None, work

    def _gather(self, local, chunks):
        import torch.distributed as dist
        return dist.gather(local, gather_list=chunks, dst=0, async_op=True)

    def finish(self):
        """Orders everything issued so far before the caller's current stream."""
        with self.torch.cuda.stream(self.comm):
            for i, w in enumerate(self.pending):
                if w is not None:
                    w.wait()
                    self.pending[i] = None
        cur = self.torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        cur.wait_stream(self.comm)


def assemble_rows(blocks, height):
    """Host-side counterpart of the de-interleave for arrays: blocks[r] holds rank r's rows
    (row r, r+G, ...), each of shape (rows, W, 3).  Returns the (height, W, 3) frame."""
    import numpy as np
    world = len(blocks)
    out = np.zeros((height,) + tuple(blocks[0].shape[1:]), blocks[0].dtype)
    for r, b in enumerate(blocks):
        n = len(range(r, height, world))
        out[r::world] = b[:n]
    return out
