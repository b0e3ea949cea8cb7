"""PoolSample ring buffer — mainGobang.jl:34-82 (duplicated in every main*.jl of the reference)."""
import numpy as np


class Sample:
    __slots__ = ("state", "policy", "player", "value", "fstate")

    def __init__(self, game):
        self.state = np.zeros(2 * game.VS, np.int8)
        self.policy = np.zeros(game.A, np.float32)
        self.player = 1
        self.value = 0.0
        self.fstate = np.zeros(game.FS, np.int8)


class PoolSample:
    """Structure-of-arrays ring buffer with the reference's semantics: 1-based write index that wraps,
    `full` flag, push_buffer / update_buffer / length_buffer."""

    def __init__(self, game, length):
        self.game, self.length = game, int(length)
        self.currentIndex, self.full = 1, False
        self.state = np.zeros((self.length, 2 * game.VS), np.int8)
        self.policy = np.zeros((self.length, game.A), np.float32)
        self.player = np.ones(self.length, np.int8)
        self.value = np.zeros(self.length, np.float32)
        self.fstate = np.zeros((self.length, game.FS), np.int8)

    def push_buffer(self, state, policy, player, i):                      # mainGobang.jl:54-68
        index = self.currentIndex
        self.state[index - 1] = state[i]
        self.policy[index - 1] = policy[i]
        self.player[index - 1] = player
        newindex = 1 if index == self.length else index + 1
        if newindex == 1:
            self.full = True
        self.currentIndex = newindex
        return index

    def update_buffer(self, index, result, fstate):                       # mainGobang.jl:70-80
        for idx in index:
            player = int(self.player[idx - 1])
            self.value[idx - 1] = (1 + result * player) / 2
            self.fstate[idx - 1] = np.asarray(fstate, np.int8) * player

    def push_generation(self, s):
        """Append a whole generation of finished samples (dict from Engine.samples(), already in PoolSample
        order) — equivalent to the reference's interleaved push_buffer/update_buffer calls."""
        n = len(s["player"])
        idx = (self.currentIndex - 1 + np.arange(n)) % self.length
        start = self.currentIndex - 1
        if n <= self.length - start:                                      # no wrap: plain slice copies (memcpy speed)
            for name in ("state", "policy", "player", "value", "fstate"):
                getattr(self, name)[start:start + n] = s[name]
        else:
            self.state[idx], self.policy[idx], self.player[idx] = s["state"], s["policy"], s["player"]
            self.value[idx], self.fstate[idx] = s["value"], s["fstate"]
        if self.currentIndex - 1 + n >= self.length:
            self.full = True
        self.currentIndex = int((self.currentIndex - 1 + n) % self.length) + 1
        return idx + 1

    def push_from_engine(self, eng):
        """The samples of the engine's last self-play generation straight into the ring (agz_get_samples unpacks the records into
        this buffer's arrays at the write position: no intermediate copies).  Same result as push_generation(eng.samples())."""
        n = eng.num_samples()
        start = self.currentIndex - 1
        if n > self.length - start:                                       # the write would wrap: the general path
            return self.push_generation(eng.samples())
        eng.samples_into(self.state[start:start + n], self.policy[start:start + n], self.player[start:start + n],
                         self.value[start:start + n], self.fstate[start:start + n])
        if start + n >= self.length:
            self.full = True
        self.currentIndex = int((start + n) % self.length) + 1
        return start + 1 + np.arange(n)

    def push_packed(self, records, n=None):
        """Packed sample records of a generation in HOST memory (agz_get_samples_packed layout, PoolSample order: uint8 array
        [n, rec_bytes] or flat) unpacked straight into the ring at the write position (agz_unpack_records: host threads, no
        device) — what a host loop calls on the pinned copy of generation k while generation k + 1 runs.  A write that reaches the
        end of the ring continues at its start (two or more contiguous segments, each unpacked in place); of more records than the
        ring holds the last `length` survive, at the positions the reference's one-by-one pushes would leave them."""
        import ctypes as C
        from . import lib as _lib
        rb = self.game.rec_bytes
        records = np.ascontiguousarray(records).reshape(-1)
        n = records.size // rb if n is None else int(n)
        L = _lib.load_library()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        first = self.currentIndex                                          # 1-based index of the first record's slot
        done = 0
        if n > self.length:                                               # the first n - length records would be overwritten by this very push
            done = n - self.length
        pos = (self.currentIndex - 1 + done) % self.length
        while done < n:
            m = min(n - done, self.length - pos)
            seg = records[done * rb:(done + m) * rb]
            rc = L.agz_unpack_records(C.byref(self.game.info), p(seg), m, p(self.state[pos:pos + m]), p(self.policy[pos:pos + m]),
                                      p(self.player[pos:pos + m]), p(self.value[pos:pos + m]), p(self.fstate[pos:pos + m]), None, None, None)
            assert rc == 0
            done += m
            pos = (pos + m) % self.length
        if self.currentIndex - 1 + n >= self.length:
            self.full = True
        self.currentIndex = int((self.currentIndex - 1 + n) % self.length) + 1
        return (first - 1 + np.arange(n)) % self.length + 1

    def length_buffer(self):                                              # mainGobang.jl:82
        return self.length if self.full else self.currentIndex - 1

    def __getitem__(self, index):
        s = Sample(self.game)
        s.state, s.policy = self.state[index - 1], self.policy[index - 1]
        s.player, s.value, s.fstate = int(self.player[index - 1]), float(self.value[index - 1]), self.fstate[index - 1]
        return s
