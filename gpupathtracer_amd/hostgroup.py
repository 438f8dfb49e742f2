"""One process per GPU without a second communication library in the process.

The reference is a single-process, single-GPU program (src/ptMain.cu:579-653); running one process
per GPU is this build's addition.  A launcher (``python -m torch.distributed.run``) starts the ranks
and exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the ranks themselves only
need (1) a way to hand RCCL's 128-byte unique id from rank 0 to the others and (2) barriers and a
max-over-ranks clock.  (1) goes through a directory on the node's local file system, (2) through
the RCCL communicator of ``libptcore.so`` itself (``pt_comm_barrier``, ``pt_comm_allreduce_max_f64``).
No other GPU runtime is loaded into the process, so the library binds to the same HIP and RCCL as
in a single-process run.

``FileGroup`` also implements barrier / max / frame-sum through files alone.  That path exists for
rehearsals on machines where RCCL cannot connect the ranks (several ranks sharing one GPU, or no
GPU at all): slow, but it exercises the same bench logic.
"""
import os
import shutil
import time

import numpy as np


def mapped_runtimes():
    """Paths of the HIP and RCCL runtimes mapped into this process (from /proc/self/maps)."""
    hip, rccl = set(), set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                fields = line.split(None, 5)
                path = fields[5].strip() if len(fields) == 6 else ""
                base = os.path.basename(path)
                if base.startswith("libamdhip64.so"):
                    hip.add(os.path.realpath(path))
                elif base.startswith("librccl.so"):
                    rccl.add(os.path.realpath(path))
    except OSError:
        pass
    return sorted(hip), sorted(rccl)


def _start_ticks(pid):
    """Field 22 of /proc/<pid>/stat (the process's start time), or "0" where that cannot be read."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def launch_key():
    """What tells one launch's ranks from another's on this node.  A launcher that names its run
    (PT_RENDEZVOUS_ID, or torch.distributed.run's TORCHELASTIC_RUN_ID when it is not the default "none")
    is taken at its word: the ranks then need not share a parent process.  Otherwise the ranks of one
    launch share the launcher as parent; the parent's start time (clock ticks since boot) keeps a
    directory a crashed launch left behind from being taken for this one's when process ids come round."""
    named = os.environ.get("PT_RENDEZVOUS_ID") or os.environ.get("TORCHELASTIC_RUN_ID")
    if named and named != "none":
        return "id_" + "".join(c if c.isalnum() or c in "-_." else "_" for c in named)[:64]
    return "ppid_%d_%s" % (os.getppid(), _start_ticks(os.getppid()))


class FileGroup:
    """Rendezvous of `world` ranks through a directory on the local file system.

    The directory is private to the user (mode 0700, ownership checked: ranks read RCCL's id and, in the
    rehearsal path, frame bytes from it).  `timeout` bounds every wait; the default (PT_RENDEZVOUS_TIMEOUT,
    60 s) is sized for ranks that start seconds apart on a cold machine, and the error names the directory
    so that a launcher whose ranks do not agree on it is found at once."""

    def __init__(self, rank, world, directory=None, timeout=None):
        if timeout is None:
            timeout = float(os.environ.get("PT_RENDEZVOUS_TIMEOUT", "60"))
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.key = launch_key()
        if directory is None:
            root = os.environ.get("PT_RENDEZVOUS_ROOT")
            if not root:
                xdg = os.environ.get("XDG_RUNTIME_DIR")           # the user's own run-time directory where there is one
                root = xdg if xdg and os.path.isdir(xdg) and os.access(xdg, os.W_OK | os.X_OK) else "/tmp"
            directory = os.path.join(root, "ptgroup_%d_%s_%s" % (os.geteuid(), os.environ.get("MASTER_PORT", "0"), self.key))
        self.dir = directory
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        st = os.stat(self.dir)
        if st.st_uid != os.geteuid() or (st.st_mode & 0o077):
            raise PermissionError("rank %d: rendezvous directory %s belongs to uid %d with mode %o; it must be this user's and private "
                                  "(set PT_RENDEZVOUS_ROOT to a directory of your own)" % (self.rank, self.dir, st.st_uid, st.st_mode & 0o777))
        self._seq = 0
        self._last_file = None
        # Files older than this are another launch's leftovers (a crashed run under a launch name that was
        # used again): they are waited past, not read.  Ranks of one launch start seconds apart.
        self._not_before = time.time() - 600.0

    # ---- small files, written atomically -------------------------------------------------------
    def _path(self, name):
        return os.path.join(self.dir, name)

    def publish(self, name, data):
        tmp = self._path(".%s.%d.tmp" % (name, self.rank))
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, self._path(name))

    def fetch(self, name):
        path = self._path(name)
        deadline = time.monotonic() + self.timeout
        delay = 0.0002
        while True:
            try:
                with open(path, "rb") as f:
                    if os.fstat(f.fileno()).st_mtime >= self._not_before:
                        return f.read()
                raise FileNotFoundError(path)                     # stale: wait for this launch's file to replace it
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise TimeoutError("rank %d of %d: %s did not appear within %.0f s.  Every rank of a launch must compute the same "
                                       "rendezvous directory (%s, keyed %s): start the ranks from one parent process, or export the same "
                                       "PT_RENDEZVOUS_ID (and MASTER_PORT) to all of them" % (self.rank, self.world, path, self.timeout, self.dir, self.key))
                time.sleep(delay)
                delay = min(delay * 1.5, 0.01)

    def broadcast(self, name, data=None, root=0):
        if self.rank == root:
            self.publish(name, data)
            return data
        return self.fetch(name)

    # ---- collectives through files (rehearsal path) --------------------------------------------
    def gather_all(self, tag, payload):
        """Every rank's payload (bytes), in rank order, on every rank."""
        self._seq += 1
        key = "%s.%d" % (tag, self._seq)
        mine = "%s.r%d" % (key, self.rank)
        self.publish(mine, payload)
        parts = [self.fetch("%s.r%d" % (key, r)) for r in range(self.world)]
        # Every rank has published this round, so every rank has finished reading the round before
        # (a rank publishes only after its own reads): this rank's previous file can go.
        if self._last_file:
            try:
                os.remove(self._path(self._last_file))
            except OSError:
                pass
        self._last_file = mine
        return parts

    def barrier(self):
        self.gather_all("barrier", b"1")

    def allreduce_max(self, value):
        return max(float(x) for x in self.gather_all("max", repr(float(value)).encode()))

    def allreduce_min_int(self, value):
        return min(int(x) for x in self.gather_all("min", str(int(value)).encode()))

    def reduce_sum(self, array, root=0):
        """Sum of the ranks' float32 arrays on `root` (added in rank order), None elsewhere."""
        a = np.ascontiguousarray(array, np.float32)
        parts = self.gather_all("sum", a.tobytes())
        if self.rank != root:
            return None
        total = np.zeros_like(a)
        for p in parts:
            total += np.frombuffer(p, np.float32).reshape(a.shape)
        return total

    def close(self):
        """Last call of every rank: rank 0 removes the directory once everybody has left."""
        self._seq += 1
        key = "bye.%d" % self._seq
        self.publish("%s.r%d" % (key, self.rank), b"1")
        if self.rank == 0:
            for r in range(self.world):
                self.fetch("%s.r%d" % (key, r))
            shutil.rmtree(self.dir, ignore_errors=True)
