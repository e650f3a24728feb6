"""Population sharding across the GPUs of one node.

One process per GPU. Rank r owns the contiguous rows [r*n/G, (r+1)*n/G) of the population /
swarm (DE: island model — donors are drawn inside the shard; the RNG is keyed by the GLOBAL row
id). Per generation ONE collective: an all-gather of each rank's best record (D+5 doubles) over
RCCL (`torch.distributed`, backend "nccl"); every rank then runs the same deterministic
finaliser on the gathered records, so the solver state (best, counters, stop flag) is identical
on all ranks without a second exchange.

torch is plumbing here (device buffers for the records, the stream, the collective); the kernels
are launched through the C-ABI on torch's current stream. Batched BFGS / LM / NM problems are
independent: they shard by simply giving every rank its own slice of the batch (no collective).

Two drivers of a turn, same results:
  * native (default on GPUs when the engine offers `comm_attach`): the engine issues the RCCL
    all-gather itself on a second stream (`nlsg_de_step_sharded`); torch.distributed only carries
    the 128-byte communicator id once. No host round trip per turn.
  * host (`NLSG_DIST_NATIVE=0`, CPU stand-ins, engines without `comm_attach`): this module orders
    turn_begin -> all_gather_into_tensor -> turn_end per turn (~85 us of host time per turn on
    the MI355X box, more than the 50 us generation it orders).
"""
import os
import sys

import numpy as np


def rccl_library_path():
    """The RCCL shared object this process already uses (PyTorch's), so that the engine's
    communicator and torch.distributed's share one library instance."""
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand if os.path.exists(cand) else ""


def shard_bounds(n, world, rank):
    """Contiguous equal split; n must divide evenly (keeps every shard's tile tree equal)."""
    if n % world:
        raise ValueError(f"population {n} is not divisible by world size {world}")
    m = n // world
    return rank * m, m


class ShardedSwarm:
    """Drives one shard per rank of a DE population or a PSO swarm.

    `engine_factory(shard_lo, shard_n, stream)` builds the rank's engine: an
    nlsolver_amd.DEEngine / PSOEngine on GPUs (tests substitute a CPU stand-in with the same
    record_doubles / turn_begin / turn_end interface)."""

    def __init__(self, dist, engine_factory, n, dim, device):
        import torch
        self.torch, self.dist = torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.lo, self.n = shard_bounds(n, self.world, self.rank)
        self.device = device
        stream = None
        if device.type == "cuda":
            stream = torch.cuda.current_stream(device).cuda_stream
        self.engine = engine_factory(self.lo, self.n, stream)
        rec = self.engine.record_doubles()
        assert rec == dim + 5
        self.send = torch.zeros(rec, dtype=torch.float64, device=device)
        self.gathered = torch.zeros(self.world * rec, dtype=torch.float64, device=device)
        # gloo with device buffers (the one-GPU rehearsal of the multi-rank flow): the collective
        # goes through host memory
        self._stage = device.type == "cuda" and dist.get_backend() == "gloo"
        if self._stage:
            self._send_host = torch.zeros(rec, dtype=torch.float64)
            self._gathered_host = torch.zeros(self.world * rec, dtype=torch.float64)
        self.native = False
        if (device.type == "cuda" and hasattr(self.engine, "comm_attach") and not self._stage
                and os.environ.get("NLSG_DIST_NATIVE", "1") != "0"):
            self._attach_native()

    def _agree(self, ok):
        """True when `ok` holds on EVERY rank (one all-reduce): what a rank does next in the
        attach sequence must not depend on its local outcome alone, or some ranks would enter a
        collective the others never reach."""
        flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item())

    def _attach_native(self):
        """One rank draws the communicator id, torch.distributed broadcasts it, every rank
        attaches (a collective inside RCCL). After each local step the ranks agree on its
        outcome; any failure anywhere puts ALL ranks on host-ordered turns."""
        from . import _capi
        import ctypes as C
        torch = self.torch

        def local(step, what):
            try:
                step()
                return True
            except RuntimeError as exc:
                print(f"nlsolver_amd.dist: rank {self.rank}: {what} failed ({exc}); "
                      "host-ordered turns instead", file=sys.stderr)
                return False

        if not self._agree(local(lambda: _capi.check(_capi.lib().nlsg_comm_load(
                rccl_library_path().encode())), "loading RCCL")):
            return
        uid = (C.c_ubyte * 128)()
        drew = self.rank != 0 or local(lambda: _capi.check(_capi.lib().nlsg_comm_unique_id(uid)),
                                       "ncclGetUniqueId")
        if not self._agree(drew):
            return
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=self.device)
        self.dist.broadcast(t, src=0)
        # the engine joins ncclCommInitRank even when its own resources failed (nlsg_comm.h), so
        # every rank comes back from this call
        attached = local(lambda: self.engine.comm_attach(bytes(t.cpu().tolist()), self.world,
                                                         self.rank), "library-side communicator")
        self.native = self._agree(attached)

    def comm_ranks(self):
        """(world, rank) read back from the library-side RCCL communicator; None when turns are
        host-ordered."""
        return self.engine.comm_ranks() if self.native else None

    def init(self, *args):
        """DE: init(x0); PSO: init(lower, upper)."""
        self.engine.init(*[np.ascontiguousarray(a, dtype=np.float64) for a in args])

    def turn(self):
        """One turn of the reference loop on the sharded population.

        DE with strategy random: the generation depends on the exchange only through the stop
        flag and is non-destructive, so it is launched while the all-gather is in flight
        (async collective on RCCL's stream) and the finaliser runs after it; a stop decided by
        the finaliser simply leaves the speculative generation unadopted. Everything else
        (strategy best, PSO: the move needs the exchanged best) keeps the serial order."""
        eng = self.engine
        eng.turn_begin(self.send.data_ptr())
        if self._stage:
            self.torch.cuda.synchronize(self.device)
            self._send_host.copy_(self.send)
            self.dist.all_gather_into_tensor(self._gathered_host, self._send_host)
            self.gathered.copy_(self._gathered_host)
            eng.turn_end(self.gathered.data_ptr(), self.world)
        elif getattr(eng, "can_speculate", lambda: False)():
            work = self.dist.all_gather_into_tensor(self.gathered, self.send, async_op=True)
            eng.turn_generation()
            work.wait()
            eng.turn_finalize(self.gathered.data_ptr(), self.world)
        else:
            self.dist.all_gather_into_tensor(self.gathered, self.send)
            eng.turn_end(self.gathered.data_ptr(), self.world)

    def step(self, turns=1):
        if self.native:
            self.engine.step_sharded(turns)
            return
        for _ in range(turns):
            self.turn()


ShardedDE = ShardedSwarm
ShardedPSO = ShardedSwarm
