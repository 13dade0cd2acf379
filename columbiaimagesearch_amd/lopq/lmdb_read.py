"""Read-only walk of an LMDB environment file (``data.mdb``) without the ``lmdb`` module.

The reference's production searcher keeps its index in LMDB (lopq/lopq/search.py:400-417: ``lmdb.open(path, max_dbs=1)``,
``open_db("index")``; keys = 4 bytes of cell + ``bytes(id)``, values = the fine codes, :455-465) and reads a cell with a cursor in key
order (:482-499).  Where py-lmdb is not installed, this module lets LOPQSearcherLMDB open such a directory for SEARCHING: it walks
the B+tree pages of the newest committed transaction and yields the named database's (key, value) pairs in key order -- what
``for key, value in txn.cursor()`` yields.

**Unpinned, restated from LMDB's published on-disk layout (lmdb.h / mdb.c of the 0.9 line) as the author remembers it**; no LMDB file
exists in the build image, so the reader is tested against the writer below only (tests/test_reference_surfaces.py).  It checks the
magic number, the data version and every page's flags, and stops with the page number at the first disagreement.

Layout relied on (64-bit, little-endian):
  * page header, 16 bytes: pgno u64, pad u16, flags u16 (BRANCH 0x01, LEAF 0x02, OVERFLOW 0x04, META 0x08), then lower u16 + upper u16
    (or, for overflow pages, the page count u32); the u16 node offsets ``mp_ptrs`` follow the header, (lower - 16) / 2 of them;
  * pages 0 and 1 are meta pages: after the header magic u32 0xBEEFC0DE, version u32 1, address u64, mapsize u64, two ``MDB_db`` records
    (free list, main) of 48 bytes -- pad u32 (the page size in the first), flags u16, depth u16, branch / leaf / overflow page counts
    u64 x 3, entries u64, root u64 -- then last page u64 and the transaction id u64: the meta page with the larger id is current;
  * node: lo u16, hi u16, flags u16 (BIGDATA 0x01, SUBDATA 0x02, DUPDATA 0x04), key size u16, key, then in a leaf the data (size
    lo | hi << 16; BIGDATA: the u64 number of the first overflow page instead) and in a branch nothing (child page = lo | hi << 16 |
    flags << 32);
  * a named database is a SUBDATA node of the main database whose data is its ``MDB_db`` record.
"""
import os
import struct

MAGIC, VERSION, HDR = 0xBEEFC0DE, 1, 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
INVALID = 0xFFFFFFFFFFFFFFFF
_DB = struct.Struct("<IHHQQQQQ")  # pad, flags, depth, branch pages, leaf pages, overflow pages, entries, root


class LMDBFormatError(ValueError):
    pass


class Env(object):
    def __init__(self, path):
        self.path = os.path.join(path, "data.mdb") if os.path.isdir(path) else path
        with open(self.path, "rb") as f:
            self.buf = f.read()
        m0 = self._meta(0)
        metas = [m0] if m0 else []
        sizes = [m0["psize"]] if m0 else [4096, 8192, 16384, 32768, 65536, 2048, 1024, 512]
        for ps in sizes:  # the second meta page sits one PAGE further
            m1 = self._meta(ps)
            if m1:
                metas.append(m1)
                break
        if not metas:
            raise LMDBFormatError("%s: no valid meta page (magic 0x%08X, version %d expected)" % (self.path, MAGIC, VERSION))
        self.psize = metas[0]["psize"]
        if self.psize < 512 or self.psize & (self.psize - 1):
            raise LMDBFormatError("%s: page size %d" % (self.path, self.psize))
        self.meta = max(metas, key=lambda m: m["txnid"])
        self.main = self.meta["main"]

    def _meta(self, off):
        if off + HDR + 136 > len(self.buf):
            return None
        flags = struct.unpack_from("<H", self.buf, off + 10)[0]
        magic, version = struct.unpack_from("<II", self.buf, off + HDR)
        if not (flags & P_META) or magic != MAGIC or version != VERSION:
            return None
        free = _DB.unpack_from(self.buf, off + HDR + 24)
        main = _DB.unpack_from(self.buf, off + HDR + 24 + 48)
        last_pg, txnid = struct.unpack_from("<QQ", self.buf, off + HDR + 24 + 96)
        return {"psize": free[0], "main": main, "last_pg": last_pg, "txnid": txnid}

    def _page(self, pgno):
        off = pgno * self.psize
        if off + self.psize > len(self.buf):
            raise LMDBFormatError("%s: page %d lies behind the end of the file" % (self.path, pgno))
        no, _pad, flags, lower, upper = struct.unpack_from("<QHHHH", self.buf, off)
        if no != pgno:
            raise LMDBFormatError("%s: page %d carries the number %d" % (self.path, pgno, no))
        return off, flags, lower, upper

    def _nodes(self, pgno, want):
        off, flags, lower, upper = self._page(pgno)
        if not (flags & want) or lower < HDR or lower > upper or upper > self.psize:
            raise LMDBFormatError("%s: page %d: flags 0x%x lower %d upper %d where a %s page was expected"
                                  % (self.path, pgno, flags, lower, upper, "branch / leaf" if want == (P_BRANCH | P_LEAF) else "leaf"))
        n = (lower - HDR) // 2
        ptrs = struct.unpack_from("<%dH" % n, self.buf, off + HDR)
        return off, flags, ptrs

    def _walk(self, root):
        """(key, value) pairs of the tree below `root` in key order."""
        if root == INVALID:
            return
        stack = [root]
        while stack:
            pgno = stack.pop()
            off, flags, ptrs = self._nodes(pgno, P_BRANCH | P_LEAF)
            if flags & P_BRANCH:
                kids = []
                for p in ptrs:
                    lo, hi, nfl, _ks = struct.unpack_from("<HHHH", self.buf, off + p)
                    kids.append(lo | (hi << 16) | (nfl << 32))
                stack.extend(reversed(kids))
                continue
            for p in ptrs:
                lo, hi, nfl, ks = struct.unpack_from("<HHHH", self.buf, off + p)
                key = bytes(self.buf[off + p + 8:off + p + 8 + ks])
                size = lo | (hi << 16)
                d0 = off + p + 8 + ks
                if nfl & F_DUPDATA:
                    raise LMDBFormatError("%s: page %d holds a sorted-duplicates node: not an index the reference writes" % (self.path, pgno))
                if nfl & F_BIGDATA:
                    opg = struct.unpack_from("<Q", self.buf, d0)[0]
                    ooff, oflags, _l, _u = self._page(opg)
                    if not (oflags & P_OVERFLOW):
                        raise LMDBFormatError("%s: page %d is not the overflow page a node of page %d names" % (self.path, opg, pgno))
                    val = bytes(self.buf[ooff + HDR:ooff + HDR + size])
                else:
                    val = bytes(self.buf[d0:d0 + size])
                yield key, val, nfl

    def db(self, name):
        """The MDB_db record of a named database (None: absent)."""
        for key, val, nfl in self._walk(self.main[7]):
            if key == name:
                if not (nfl & F_SUBDATA) or len(val) != 48:
                    raise LMDBFormatError("%s: %r is not a named database" % (self.path, name))
                return _DB.unpack(val)
        return None

    def items(self, name=b"index"):
        """(key, value) of the named database in key order -- what a cursor over it yields."""
        rec = self.db(name)
        if rec is None:
            return
        for key, val, _ in self._walk(rec[7]):
            yield key, val

    def entries(self, name=b"index"):
        rec = self.db(name)
        return 0 if rec is None else rec[6]


# ---- the writer: the same statement of the layout run backwards (tests; documents the expected file) ---------------------------------
def write_env(path, items, name=b"index", psize=4096, txnid=7):
    """A data.mdb whose database `name` holds `items` ((key, value) byte pairs; sorted here), values above a quarter page on overflow pages."""
    items = sorted(items)
    pages = {}  # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        pg = next_pg[0]
        next_pg[0] += n
        return pg

    def build(flags, nodes):  # nodes: list of bytes, in order; returns the page image without the number
        ptrs, body, upper = [], {}, psize
        for nd in nodes:
            nd = nd + (b"\0" if len(nd) & 1 else b"")
            upper -= len(nd)
            ptrs.append(upper)
            body[upper] = nd
        lower = HDR + 2 * len(nodes)
        assert lower <= upper, "page overflow"
        img = bytearray(psize)
        struct.pack_into("<HHHH", img, 8, 0, flags, lower, upper)
        struct.pack_into("<%dH" % len(ptrs), img, HDR, *ptrs)
        for o, nd in body.items():
            img[o:o + len(nd)] = nd
        return img

    def finish(pg, img):
        struct.pack_into("<Q", img, 0, pg)
        pages[pg] = bytes(img)

    def leaf_node(key, val, nfl=0):
        if nfl == 0 and len(val) > psize // 4:
            npg = (HDR + len(val) + psize - 1) // psize
            opg = alloc(npg)
            img = bytearray(npg * psize)
            struct.pack_into("<QHHI", img, 0, opg, 0, P_OVERFLOW, npg)
            img[HDR:HDR + len(val)] = val
            for i in range(npg):
                pages[opg + i] = bytes(img[i * psize:(i + 1) * psize])
            return struct.pack("<HHHH", len(val) & 0xFFFF, len(val) >> 16, F_BIGDATA, len(key)) + key + struct.pack("<Q", opg)
        return struct.pack("<HHHH", len(val) & 0xFFFF, len(val) >> 16, nfl, len(key)) + key + val

    def tree(entries, counts):
        """Pages of a tree over sorted (key, node bytes); returns (root, depth)."""
        if not entries:
            return INVALID, 0
        level, depth = [], 1
        cur, used, first = [], HDR, None
        for key, nd in entries:
            need = len(nd) + (len(nd) & 1) + 2
            if cur and used + need > psize:
                pg = alloc(); finish(pg, build(P_LEAF, cur)); level.append((first, pg)); counts["leaf"] += 1
                cur, used, first = [], HDR, None
            if first is None:
                first = key
            cur.append(nd); used += need
        pg = alloc(); finish(pg, build(P_LEAF, cur)); level.append((first, pg)); counts["leaf"] += 1
        while len(level) > 1:
            nxt, cur, used, first = [], [], HDR, None
            for i, (key, child) in enumerate(level):
                k = b"" if not cur else key  # the leftmost key of a branch page is not stored
                nd = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, child >> 32, len(k)) + k
                need = len(nd) + (len(nd) & 1) + 2
                if cur and used + need > psize:
                    pg = alloc(); finish(pg, build(P_BRANCH, cur)); nxt.append((first, pg)); counts["branch"] += 1
                    cur, used, first = [], HDR, None
                    nd = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, child >> 32, 0)
                if first is None:
                    first = key
                cur.append(nd); used += len(nd) + (len(nd) & 1) + 2
            pg = alloc(); finish(pg, build(P_BRANCH, cur)); nxt.append((first, pg)); counts["branch"] += 1
            level, depth = nxt, depth + 1
        return level[0][1], depth

    counts = {"leaf": 0, "branch": 0}
    root, depth = tree([(k, leaf_node(k, v)) for k, v in items], counts)
    sub = _DB.pack(0, 0, depth, counts["branch"], counts["leaf"], 0, len(items), root)
    mcounts = {"leaf": 0, "branch": 0}
    mroot, mdepth = tree([(name, leaf_node(name, sub, F_SUBDATA))], mcounts)
    last = next_pg[0] - 1
    for pg, tid in ((0, txnid - 1), (1, txnid)):  # the older meta page names an empty main database: the reader must pick the newer
        img = bytearray(psize)
        struct.pack_into("<QHH", img, 0, pg, 0, P_META)
        struct.pack_into("<IIQQ", img, HDR, MAGIC, VERSION, 0, 1 << 30)
        img[HDR + 24:HDR + 72] = _DB.pack(psize, 0, 0, 0, 0, 0, 0, INVALID)
        img[HDR + 72:HDR + 120] = _DB.pack(0, 0, mdepth, 0, 1, 0, 1, mroot) if pg == 1 else _DB.pack(0, 0, 0, 0, 0, 0, 0, INVALID)
        struct.pack_into("<QQ", img, HDR + 120, last, tid)
        pages[pg] = bytes(img)
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "data.mdb"), "wb") as f:
        for pg in range(next_pg[0]):
            f.write(pages[pg])
    return counts, depth
