"""The reference's persistence layout: the `hnswindex` and `hnswnodet` module types stream
their fields through RedisModule_Save{String,Unsigned,Double,Float} in a fixed order
(src/types.rs:243-284 save_index / :180-241 load_index, :410-428 save_node / :373-408 load_node), and
src/lib.rs:252-315 (make_index) rebuilds an Index from one IndexRedis value plus one NodeRedis value per
node.  This module restates exactly that: the two field sequences, the two conversions
(IndexRedis::from / NodeRedis::from, src/types.rs:62-91, 292-309) and make_index -- which here ends in ONE
hnsw_import call, so an existing dataset goes straight into HBM.

Byte container.  Redis wraps every Save* call of a module value in an opcode (rdb.h, "module value
format 2": RDB_MODULE_OPCODE_EOF=0, SINT=1, UINT=2, FLOAT=3, DOUBLE=4, STRING=5), integers and lengths in
the RDB length encoding, floats/doubles as little-endian IEEE bytes, the value closed by an EOF opcode.
ModuleIO below writes and reads that container.  It WRITES strings as raw length + bytes (always accepted);
it READS the special encodings redis-server may choose when it writes a string -- 8/16/32-bit integers
(0xC0..0xC2) and LZF-compressed bytes (0xC3) -- as well.  redis-server and the redis-module crate are absent
from this image, so the container
is restated from Redis's published format and is not byte-pinned against a real dump; what the parity
tests pin is the FIELD ORDER and the graph semantics, which is all the engine depends on.
"""
import struct

import numpy as np

OP_EOF, OP_SINT, OP_UINT, OP_FLOAT, OP_DOUBLE, OP_STRING = 0, 1, 2, 3, 4, 5
PREFIX = "hnsw"                      # src/lib.rs:27


class RdbFormatError(ValueError):
    pass


def _len_encode(n):
    """rdbSaveLen: 6-bit, 14-bit, 32-bit or 64-bit big-endian forms"""
    if n < 0:
        raise RdbFormatError("negative length")
    if n < (1 << 6):
        return bytes([n])
    if n < (1 << 14):
        return bytes([0x40 | (n >> 8), n & 0xFF])
    if n <= 0xFFFFFFFF:
        return b"\x80" + struct.pack(">I", n)
    return b"\x81" + struct.pack(">Q", n)


def lzf_decompress(data, out_len):
    """liblzf's format, as rdbLoadLzfStringObject feeds it: literal runs (ctrl < 32: ctrl + 1 bytes follow) and
    back references (length ctrl >> 5, +next byte when 7, then +2; offset ((ctrl & 0x1f) << 8 | next) + 1)"""
    out = bytearray()
    i, n = 0, len(data)
    while i < n:
        ctrl = data[i]
        i += 1
        if ctrl < 32:
            run = ctrl + 1
            if i + run > n:
                raise RdbFormatError("truncated LZF literal run")
            out += data[i:i + run]
            i += run
        else:
            length = ctrl >> 5
            if length == 7:
                if i >= n:
                    raise RdbFormatError("truncated LZF back reference")
                length += data[i]
                i += 1
            if i >= n:
                raise RdbFormatError("truncated LZF back reference")
            ref = len(out) - (((ctrl & 0x1F) << 8) | data[i]) - 1
            i += 1
            if ref < 0:
                raise RdbFormatError("LZF back reference before the start of the output")
            for _ in range(length + 2):
                out.append(out[ref])
                ref += 1
    if len(out) != out_len:
        raise RdbFormatError("LZF stream decodes to %d bytes, header says %d" % (len(out), out_len))
    return bytes(out)


def index_key(name):
    """the Redis key of an index: 'hnsw.{idx}' (src/lib.rs:27, 137); full keys pass through"""
    return name if name.startswith(PREFIX + ".") else "%s.%s" % (PREFIX, name)


def node_key(index_name, node):
    """the Redis key of a node: 'hnsw.{idx}.{node}' (src/lib.rs:342-343); full keys pass through"""
    return node if node.startswith(PREFIX + ".") else "%s.%s" % (index_key(index_name), node)


class ModuleIO:
    """RedisModuleIO stand-in: the Save*/Load* calls of redismodule.h over a byte buffer."""

    def __init__(self, data=b""):
        self._out = bytearray()
        self._in = memoryview(bytes(data))
        self._pos = 0

    # ---- writer ---------------------------------------------------------------------
    def save_unsigned(self, v):
        self._out += _len_encode(OP_UINT) + _len_encode(int(v))

    def save_string(self, s):
        b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
        self._out += _len_encode(OP_STRING) + _len_encode(len(b)) + b

    def save_double(self, v):
        self._out += _len_encode(OP_DOUBLE) + struct.pack("<d", float(v))

    def save_float(self, v):
        self._out += _len_encode(OP_FLOAT) + struct.pack("<f", float(v))

    def finish(self):
        """the value's closing EOF opcode; returns the bytes"""
        self._out += _len_encode(OP_EOF)
        return bytes(self._out)

    # ---- reader ---------------------------------------------------------------------
    def _take(self, n):
        if self._pos + n > len(self._in):
            raise RdbFormatError("truncated module value")
        b = self._in[self._pos:self._pos + n]
        self._pos += n
        return b

    def _load_len(self):
        b0 = self._take(1)[0]
        kind = b0 >> 6
        if kind == 0:
            return b0 & 0x3F
        if kind == 1:
            return ((b0 & 0x3F) << 8) | self._take(1)[0]
        if b0 == 0x80:
            return struct.unpack(">I", self._take(4))[0]
        if b0 == 0x81:
            return struct.unpack(">Q", self._take(8))[0]
        raise RdbFormatError("special string encodings are not raw lengths (byte 0x%02x)" % b0)

    def _expect(self, op):
        got = self._load_len()
        if got != op:
            raise RdbFormatError("expected module opcode %d, found %d" % (op, got))

    def load_unsigned(self):
        self._expect(OP_UINT)
        return self._load_len()

    def load_string(self):
        """rdbLoadStringObject: a raw length + bytes, or one of the special encodings (first byte 0b11xxxxxx)"""
        self._expect(OP_STRING)
        b0 = self._in[self._pos] if self._pos < len(self._in) else None
        if b0 is not None and (b0 >> 6) == 3:
            self._pos += 1
            enc = b0 & 0x3F
            if enc == 0:                                         # RDB_ENC_INT8
                return str(struct.unpack("<b", self._take(1))[0])
            if enc == 1:                                         # RDB_ENC_INT16
                return str(struct.unpack("<h", self._take(2))[0])
            if enc == 2:                                         # RDB_ENC_INT32
                return str(struct.unpack("<i", self._take(4))[0])
            if enc == 3:                                         # RDB_ENC_LZF: compressed length, original length, bytes
                clen = self._load_len()
                ulen = self._load_len()
                return lzf_decompress(bytes(self._take(clen)), ulen).decode("utf-8")
            raise RdbFormatError("unknown string encoding 0x%02x" % b0)
        n = self._load_len()
        return bytes(self._take(n)).decode("utf-8")

    def load_double(self):
        self._expect(OP_DOUBLE)
        return struct.unpack("<d", self._take(8))[0]

    def load_float(self):
        self._expect(OP_FLOAT)
        return struct.unpack("<f", self._take(4))[0]

    def expect_eof(self):
        self._expect(OP_EOF)
        if self._pos != len(self._in):
            raise RdbFormatError("bytes after the value's EOF opcode")


class IndexRedis:
    """src/types.rs:45-60"""

    def __init__(self, name="", mfunc_kind="Euclidean", data_dim=0, m=0, m_max=0, m_max_0=0, ef_construction=0,
                 level_mult=0.0, node_count=0, max_layer=0, layers=None, nodes=None, enterpoint=None):
        self.name, self.mfunc_kind, self.data_dim, self.m, self.m_max, self.m_max_0 = name, mfunc_kind, data_dim, m, m_max, m_max_0
        self.ef_construction, self.level_mult, self.node_count, self.max_layer = ef_construction, level_mult, node_count, max_layer
        self.layers = layers if layers is not None else []      # names of the nodes whose TOP layer is l (core.rs:596)
        self.nodes = nodes if nodes is not None else []          # every node key
        self.enterpoint = enterpoint                             # node key or None

    def __eq__(self, o):
        return isinstance(o, IndexRedis) and self.__dict__ == o.__dict__


class NodeRedis:
    """src/types.rs:286-290"""

    def __init__(self, data=None, neighbors=None):
        self.data = [float(np.float32(x)) for x in (data if data is not None else [])]
        self.neighbors = neighbors if neighbors is not None else []   # per layer, names in stored order

    def __eq__(self, o):
        return isinstance(o, NodeRedis) and self.data == o.data and self.neighbors == o.neighbors


def save_index(ir):
    """src/types.rs:243-284, field for field"""
    io = ModuleIO()
    io.save_string(ir.name)                     # :248-249
    io.save_string(ir.mfunc_kind)               # :251-252
    io.save_unsigned(ir.data_dim)               # :254
    io.save_unsigned(ir.m)                      # :255
    io.save_unsigned(ir.m_max)                  # :256
    io.save_unsigned(ir.m_max_0)                # :257
    io.save_unsigned(ir.ef_construction)        # :258
    io.save_double(ir.level_mult)               # :259
    io.save_unsigned(ir.node_count)             # :260
    io.save_unsigned(ir.max_layer)              # :261
    io.save_unsigned(len(ir.layers))            # :263
    for layer in ir.layers:                     # :264-270
        io.save_unsigned(len(layer))
        for n in layer:
            io.save_string(n)
    io.save_unsigned(len(ir.nodes))             # :272
    for n in ir.nodes:                          # :273-276
        io.save_string(n)
    io.save_string(ir.enterpoint if ir.enterpoint is not None else "null")   # :278-283
    return io.finish()


def load_index(blob):
    """src/types.rs:180-241"""
    io = ModuleIO(blob)
    ir = IndexRedis()
    ir.name = io.load_string()
    ir.mfunc_kind = io.load_string()
    ir.data_dim = io.load_unsigned()
    ir.m = io.load_unsigned()
    ir.m_max = io.load_unsigned()
    ir.m_max_0 = io.load_unsigned()
    ir.ef_construction = io.load_unsigned()
    ir.level_mult = io.load_double()
    ir.node_count = io.load_unsigned()
    ir.max_layer = io.load_unsigned()
    ir.layers = []
    for _ in range(io.load_unsigned()):
        ir.layers.append([io.load_string() for _ in range(io.load_unsigned())])
    ir.nodes = [io.load_string() for _ in range(io.load_unsigned())]
    ep = io.load_string()
    ir.enterpoint = None if ep == "null" else ep          # :233-236
    io.expect_eof()
    return ir


def save_node(nr):
    """src/types.rs:410-428"""
    io = ModuleIO()
    io.save_unsigned(len(nr.data))              # :415
    for d in nr.data:                           # :416-418
        io.save_float(d)
    io.save_unsigned(len(nr.neighbors))         # :420
    for layer in nr.neighbors:                  # :421-427
        io.save_unsigned(len(layer))
        for n in layer:
            io.save_string(n)
    return io.finish()


def load_node(blob):
    """src/types.rs:373-408"""
    io = ModuleIO(blob)
    nr = NodeRedis()
    nr.data = [io.load_float() for _ in range(io.load_unsigned())]
    nr.neighbors = []
    for _ in range(io.load_unsigned()):
        nr.neighbors.append([io.load_string() for _ in range(io.load_unsigned())])
    io.expect_eof()
    return nr


# ---- conversions ------------------------------------------------------------------------------
def graph_to_redis(name, dim, m, ef_construction, graph, names, dead=None):
    """IndexRedis::from + NodeRedis::from (src/types.rs:62-91, 292-309) for a graph in the engine's export
    form (levels, enterpoint, per-layer CSR, vectors; see Index.export_graph) whose node i has key names[i].
    Tombstoned ids (dead[i]) are left out, as the reference drops deleted nodes (core.rs:419).  The reference
    walks a HashMap / HashSet here (arbitrary order); this restatement emits ascending ids."""
    n = len(graph["levels"])
    live = [i for i in range(n) if not (dead is not None and dead[i])]
    levels = graph["levels"]
    max_layer = int(graph["max_layer"])
    ir = IndexRedis(name=name, mfunc_kind="Euclidean", data_dim=dim, m=m, m_max=m, m_max_0=2 * m,
                    ef_construction=ef_construction, level_mult=1.0 / float(np.log(float(m))),
                    node_count=len(live), max_layer=max_layer,
                    layers=[[names[i] for i in live if int(levels[i]) == l] for l in range(max_layer + 1)] if live else [],
                    nodes=[names[i] for i in live],
                    enterpoint=names[int(graph["enterpoint"])] if live and int(graph["enterpoint"]) >= 0 else None)
    nodes = {}
    V = graph["vectors"]
    for i in live:
        rows = []
        for l in range(min(int(levels[i]), max_layer) + 1):
            rp, cl = graph["row_ptr"][l], graph["col"][l]
            rows.append([names[int(j)] for j in cl[int(rp[i]):int(rp[i + 1])]])
        nodes[names[i]] = NodeRedis(data=V[i], neighbors=rows)
    return ir, nodes


def redis_to_graph(ir, get_node):
    """make_index (src/lib.rs:252-315): node keys -> dense ids in the order of ir.nodes, neighbour names ->
    ids (an unknown name is the reference's "Node: {} does not exist" error), levels from the layer sets
    (core.rs:596: a node belongs to the set of its top layer).  Returns (graph, names)."""
    names = list(ir.nodes)
    ids = {nm: i for i, nm in enumerate(names)}
    if len(ids) != len(names):
        raise RdbFormatError("duplicate node key in the index value")
    n = len(names)
    levels = np.zeros(n, dtype=np.uint32)
    seen = np.zeros(n, dtype=bool)
    for l, layer in enumerate(ir.layers):
        for nm in layer:
            if nm not in ids:
                raise KeyError("Node: %s does not exist" % nm)        # src/lib.rs:293
            levels[ids[nm]] = l
            seen[ids[nm]] = True
    if n and not seen.all():
        raise RdbFormatError("a node is in no layer set")
    L = len(ir.layers)
    deg = [np.zeros(n, dtype=np.uint64) for _ in range(L)]
    cols = [[] for _ in range(L)]
    V = np.zeros((n, ir.data_dim), dtype=np.float32)
    rows_of = []
    for i, nm in enumerate(names):
        nr = get_node(nm)
        if nr is None:
            raise KeyError("Node: %s does not exist" % nm)            # src/lib.rs:262
        if len(nr.data) != ir.data_dim:
            raise RdbFormatError("node %s has %d components, the index %d" % (nm, len(nr.data), ir.data_dim))
        V[i] = np.asarray(nr.data, dtype=np.float32)
        rows_of.append(nr.neighbors)
    for l in range(L):
        for i in range(n):
            row = rows_of[i][l] if l < len(rows_of[i]) else []
            if row and l > levels[i]:
                raise RdbFormatError("node %s has links above its layer" % names[i])
            for nb in row:
                if nb not in ids:
                    raise KeyError("Node: %s does not exist" % nb)    # src/lib.rs:279
                cols[l].append(ids[nb])
            deg[l][i] = len(row)
    row_ptr = []
    for l in range(L):
        rp = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(deg[l], out=rp[1:])
        row_ptr.append(rp)
    ep = -1 if ir.enterpoint is None else ids.get(ir.enterpoint, None)
    if ep is None:
        raise KeyError("Node: %s does not exist" % ir.enterpoint)     # src/lib.rs:307
    graph = dict(vectors=V, levels=levels, enterpoint=ep, max_layer=int(ir.max_layer), row_ptr=row_ptr,
                 col=[np.asarray(c, dtype=np.uint32) for c in cols])
    return graph, names


def dump_index(index, qualify=True):
    """An engine Index -> (hnswindex value bytes, {node key: hnswnodet value bytes}): what the module's
    RDB save callbacks stream for this index and each of its nodes.  The reference stores FULL keys -- the index
    as 'hnsw.{idx}', nodes as 'hnsw.{idx}.{node}' -- and its make_index opens the stored names as keys
    (src/lib.rs:256-258), so names that are not full keys yet are qualified (qualify=False keeps them verbatim)."""
    g = index.export_graph(with_vectors=True)
    dead = [nm is None for nm in index._names]
    iname = index_key(index.name) if qualify else index.name
    names = [(node_key(index.name, nm) if qualify else nm) if nm is not None else "" for nm in index._names]
    ir, nodes = graph_to_redis(iname, index.data_dim, index.m, index.ef_construction, g, names, dead)
    return save_index(ir), {k: save_node(v) for k, v in nodes.items()}


def restore_index(index_blob, node_blobs, device=0, seed=0):
    """make_index over the two value types: -> a new engine Index holding the same graph (one hnsw_import)."""
    from .index import Index
    ir = load_index(index_blob)
    graph, names = redis_to_graph(ir, lambda nm: load_node(node_blobs[nm]) if nm in node_blobs else None)
    idx = Index(ir.name, ir.data_dim, ir.m, ir.ef_construction, seed=seed, device=device)
    if names:
        idx.import_graph(graph, names=names)
    return idx
