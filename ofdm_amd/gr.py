"""The few GNU Radio runtime objects the reference's transmit scripts touch (messages, the
bounded message queue, sinks, multiply_const, top_block), reduced to what a batch modulator
needs.  There is no scheduler: a top_block records how the transmit path is wired to its sink
and the path pushes finished IQ batches downstream.

Only the call surface used by the reference's ftw_ofdm.py / ftw_ofdm_tx_uhd.py is provided.
"""
import collections
import threading

import numpy as np

sizeof_gr_complex = 8
sizeof_char = 1
sizeof_float = 4
RT_OK = 0


def enable_realtime_scheduling():
    """Nothing to schedule: kernels run on a HIP stream."""
    return RT_OK


class message(object):
    """gr.message: type 1 marks end of stream (examples/ftw_ofdm.py:199)."""

    def __init__(self, type=0, arg1=0.0, arg2=0.0, data=b""):
        self._type, self._arg1, self._arg2, self._data = type, arg1, arg2, bytes(data)

    def type(self):
        return self._type

    def to_string(self):
        return self._data

    def length(self):
        return len(self._data)


def message_from_string(s, type=0, arg1=0.0, arg2=0.0):
    return message(type, arg1, arg2, s if isinstance(s, (bytes, bytearray)) else s.encode("latin-1"))


class msg_queue(object):
    """gr.msg_queue(limit): insert_tail blocks while `limit` messages wait (0 = unbounded)."""

    def __init__(self, limit=0):
        self._limit = limit
        self._q = collections.deque()
        self._cv = threading.Condition()

    def limit(self):
        return self._limit

    def count(self):
        with self._cv:
            return len(self._q)

    def empty_p(self):
        return self.count() == 0

    def full_p(self):
        return self._limit != 0 and self.count() >= self._limit

    def insert_tail(self, msg):
        with self._cv:
            while self._limit and len(self._q) >= self._limit:
                self._cv.wait()
            self._q.append(msg)
            self._cv.notify_all()

    def delete_head(self):
        with self._cv:
            while not self._q:
                self._cv.wait()
            m = self._q.popleft()
            self._cv.notify_all()
            return m

    def delete_head_nowait(self):
        with self._cv:
            if not self._q:
                return None
            m = self._q.popleft()
            self._cv.notify_all()
            return m

    def flush(self):
        with self._cv:
            self._q.clear()
            self._cv.notify_all()


def _to_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x)


class _PinnedStager(object):
    """Device -> host in chunks through two page-locked buffers on a copy stream, so that the copy
    of chunk i+1 overlaps whatever the consumer does with chunk i (writing it to a file).  A
    pageable `.cpu()` of the same data runs at a fraction of the link rate."""

    def __init__(self, chunk_bytes=128 << 20):
        self._chunk = int(chunk_bytes)
        self._bufs = None
        self._stream = None

    def chunks(self, t):
        """Yields numpy views (valid until the next iteration) covering tensor t in order."""
        import torch
        flat = t.detach().contiguous().view(torch.uint8).reshape(-1)
        if not flat.is_cuda:
            yield flat.numpy()
            return
        if self._bufs is None:
            self._bufs = [torch.empty(self._chunk, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            self._stream = torch.cuda.Stream(device=flat.device)
        n = flat.numel()
        self._stream.wait_stream(torch.cuda.current_stream(flat.device))     # the kernels that produced t
        events = [None, None]

        def start(i):
            lo = i * self._chunk
            hi = min(n, lo + self._chunk)
            with torch.cuda.stream(self._stream):
                self._bufs[i & 1][:hi - lo].copy_(flat[lo:hi], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            events[i & 1] = ev
            return hi - lo

        n_chunks = (n + self._chunk - 1) // self._chunk
        size = start(0) if n_chunks else 0
        for i in range(n_chunks):
            nxt = start(i + 1) if i + 1 < n_chunks else 0
            events[i & 1].synchronize()
            yield self._bufs[i & 1][:size].numpy()
            size = nxt
        flat.record_stream(self._stream)


class file_sink(object):
    """gr.file_sink(itemsize, filename): raw little-endian items appended to a file
    (examples/ftw_ofdm_tx_uhd.py:66-67: complex64 = float32 I,Q pairs)."""

    def __init__(self, itemsize, filename):
        self.itemsize = itemsize
        self.filename = filename
        self._f = open(filename, "wb")
        self._stager = _PinnedStager()

    def write(self, items):
        if hasattr(items, "detach"):
            for part in self._stager.chunks(items):
                part.tofile(self._f)
        else:
            _to_numpy(items).tofile(self._f)

    def close(self):
        if self._f:
            self._f.close()
            self._f = None


class file_source(object):
    """gr.file_source(itemsize, filename): replays raw complex64 samples
    (examples/ftw_ofdm_tx_uhd.py:75-78) through the connected multiply_const_cc into the sinks, in
    chunks, on the GPU."""

    def __init__(self, itemsize, filename, repeat=False, chunk_items=1 << 24, device=None):
        if itemsize != sizeof_gr_complex:
            raise ValueError("only complex64 streams are replayed")
        self.filename, self._chunk, self._device = filename, int(chunk_items), device
        self._outs = []              # (multiply_const_cc or None, sink) per connection

    def _attach(self, gain, sinks):
        self._outs.extend((gain, s) for s in sinks)

    def _start(self):
        pass

    def _finish(self):
        import ctypes as C
        import torch
        from . import _lib
        from .ftw_packet_utils import _context
        ctx = _context(self._device)
        with open(self.filename, "rb") as f:
            while True:
                host = np.fromfile(f, dtype=np.complex64, count=self._chunk)
                if host.size == 0:
                    break
                dev = torch.from_numpy(host).to(ctx.device)
                stream = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
                for gain, sink in self._outs:
                    out = dev
                    if gain is not None:
                        k = complex(gain.k())
                        out = torch.empty_like(dev)
                        _lib.check(ctx._lib.ofdmtx_scale_cc(ctx._h, C.c_void_p(dev.data_ptr()), dev.numel(), k.real,
                                                            k.imag, C.c_void_p(out.data_ptr()), stream))
                    sink.write(out)


class null_sink(object):
    def __init__(self, itemsize):
        self.itemsize = itemsize
        self.items_seen = 0

    def write(self, items):
        self.items_seen += int(items.numel() if hasattr(items, "numel") else np.size(items))

    def close(self):
        pass


class vector_sink_c(object):
    """Keeps everything it receives (host memory)."""

    def __init__(self, vlen=1):
        self._chunks = []

    def write(self, items):
        self._chunks.append(_to_numpy(items).reshape(-1))

    def data(self):
        return np.concatenate(self._chunks) if self._chunks else np.zeros(0, np.complex64)

    def close(self):
        pass


class vector_source_c(object):
    """gr.vector_source_c(data, repeat, vlen): items of vlen complex values; with repeat the data
    starts over when it runs out (rawofdm/examples/ofdm_rxtx.py:120)."""

    def __init__(self, data, repeat=False, vlen=1):
        self.vlen = int(vlen)
        d = np.ascontiguousarray(np.asarray(data, dtype=np.complex64).reshape(-1))
        if d.size % self.vlen:
            raise ValueError("vector_source_c: data is not a whole number of items")
        self._data, self._repeat, self._pos, self._dev = d.reshape(-1, self.vlen), bool(repeat), 0, None

    def take(self, n_items, device=None):
        """The next n_items items (fewer at the end of a non-repeating source) as a flat tensor."""
        import torch
        n_have = self._data.shape[0]
        if not self._repeat:
            n_items = min(int(n_items), n_have - self._pos)
        if n_have == 0 or n_items <= 0:
            return torch.zeros(0, dtype=torch.complex64, device=device)
        if self._dev is None or (device is not None and self._dev.device != torch.device(device)):
            self._dev = torch.from_numpy(self._data).to(device if device is not None else "cpu")
        idx = (torch.arange(self._pos, self._pos + n_items, device=self._dev.device) % n_have)
        self._pos = (self._pos + n_items) % n_have if self._repeat else self._pos + n_items
        return self._dev.index_select(0, idx).reshape(-1)


class multiply_const_cc(object):
    def __init__(self, k):
        self._k = k

    def k(self):
        return self._k

    def set_k(self, k):
        self._k = k


class hier_block2(object):
    def __init__(self, name="", input_signature=None, output_signature=None):
        self._name = name

    def name(self):
        return self._name


class top_block(hier_block2):
    """connect(source, [multiply_const_cc], sink...) wires a transmit path; start()/wait()/run()
    keep the reference's control flow (examples/ftw_ofdm_tx_uhd.py:155-175)."""

    def __init__(self, name="top_block"):
        hier_block2.__init__(self, name)
        self._sources = []
        self._sinks = []

    def connect(self, *blocks):
        src = blocks[0]
        if hasattr(src, "take") and len(blocks) == 2 and hasattr(blocks[1], "_set_source"):
            blocks[1]._set_source(src)           # data source -> transmit path (rawofdm/examples/ofdm_tx.py:51)
            return
        if not hasattr(src, "_attach"):
            raise TypeError("the first block of a connection must be a transmit path")
        gain = None
        sinks = []
        for b in blocks[1:]:
            if isinstance(b, multiply_const_cc):
                gain = b
            elif hasattr(b, "write"):
                sinks.append(b)
            else:
                raise TypeError("cannot connect %r" % (b,))
        src._attach(gain, sinks)
        if src not in self._sources:
            self._sources.append(src)
        self._sinks.extend(sinks)

    def start(self):
        for s in self._sources:
            s._start()

    def wait(self):
        for s in self._sources:
            s._finish()
        for k in self._sinks:
            k.close()

    def stop(self):
        pass

    def run(self):
        self.start()
        self.wait()
