"""Persistent key/value store for LOPQSearcherLMDB when the `lmdb` module is not installed.

The reference keeps its production index in LMDB (lopq/lopq/search.py:416-417): key = cell (2 x uint16,
``array('H')``) + ``bytes(id)``, value = the fine codes as bytes, ``put`` replaces an existing key (:445-470), a cell is
read back with a cursor in key order (:482-499), and one ``add_codes`` call is ONE write transaction (:445, :467) -- it is
stored as a whole or not at all.  This file keeps exactly those keys and values -- same bytes, same last-write-wins rule,
same all-or-nothing unit -- in an append-only log, so that an index survives a restart without LMDB:

    <lmdb_path>/index.ciskv :=  b"CISKV2\n"  group*
    group  :=  u32le n_records, u32le payload_bytes, u32le crc32(payload), payload
    payload := { u32le key_len, u32le val_len, key, value } * n_records

A group is one ``append`` call (= one add_codes call = one LMDB transaction).  On open a group that runs past the end of
the file (a crash in the middle of the write) is dropped AS A WHOLE and the file is cut back to the last complete group;
a complete group whose checksum or record framing is wrong is corruption in the middle of the data and raises instead of
silently dropping everything behind it.  ``compact`` rewrites the log with one record per live key.  Logs of the first
format (``CISKV1``: bare records, no groups) are read and rewritten in this one on open.
It is NOT the LMDB file format: a deployment that has LMDB files keeps using them through the `lmdb` module
(LOPQSearcherLMDB refuses to shadow an existing ``data.mdb`` with a log).
"""
import os
import struct
import zlib

MAGIC = b"CISKV2\n"
MAGIC_V1 = b"CISKV1\n"
FILE_NAME = "index.ciskv"


def _pack_records(items):
    buf = bytearray()
    k = 0
    for key, val in items:
        buf += struct.pack("<II", len(key), len(val))
        buf += key
        buf += val
        k += 1
    return buf, k


def _unpack_records(data, pos, end, n_expected=None):
    """Records of data[pos:end]; returns (records, position after the last complete record)."""
    out = []
    while pos + 8 <= end:
        kl, vl = struct.unpack_from("<II", data, pos)
        if pos + 8 + kl + vl > end:
            break
        out.append((data[pos + 8:pos + 8 + kl], data[pos + 8 + kl:pos + 8 + kl + vl]))
        pos += 8 + kl + vl
        if n_expected is not None and len(out) == n_expected:
            break
    return out, pos


class KVLog(object):
    def __init__(self, path):
        self.dir = path
        os.makedirs(path, exist_ok=True)
        self.path = os.path.join(path, FILE_NAME)
        self.records = 0
        self.dropped_torn_bytes = 0  # what the last load() cut off the tail (a torn group)
        if not os.path.exists(self.path):
            with open(self.path, "wb") as f:
                f.write(MAGIC)
                f.flush()
                os.fsync(f.fileno())

    def load(self):
        """(key, value) pairs in log order.  A torn last group is dropped as a whole (and cut off the file); damage in the
        middle of the log raises ValueError."""
        with open(self.path, "rb") as f:
            data = f.read()
        n = len(data)
        if data[:len(MAGIC_V1)] == MAGIC_V1:  # first format: bare records, a torn tail record is cut off
            out, good = _unpack_records(data, len(MAGIC_V1), n)
            self.dropped_torn_bytes = n - good
            self.compact(out)
            return out
        if data[:len(MAGIC)] != MAGIC:
            raise ValueError("%s is not a CISKV log" % self.path)
        pos = good = len(MAGIC)
        out = []
        while pos < n:
            if pos + 12 > n:
                break  # torn group header
            nrec, nbytes, crc = struct.unpack_from("<III", data, pos)
            if pos + 12 + nbytes > n:
                break  # torn group: the write of this add_codes call never completed -> dropped as a whole
            payload = data[pos + 12:pos + 12 + nbytes]
            recs, end = _unpack_records(payload, 0, nbytes, nrec) if nrec else ([], 0)
            if (zlib.crc32(payload) & 0xFFFFFFFF) != crc or len(recs) != nrec or end != nbytes:
                raise ValueError("%s: group at byte %d is damaged (checksum / framing); %d records before it are intact -- "
                                 "refusing to drop the %d bytes behind it silently" % (self.path, pos, len(out), n - pos))
            out.extend(recs)
            pos += 12 + nbytes
            good = pos
        self.dropped_torn_bytes = n - good
        if good != n:
            with open(self.path, "r+b") as f:
                f.truncate(good)
        self.records = len(out)
        return out

    def append(self, items):
        """items: iterable of (key bytes, value bytes) of ONE add_codes call: one group, one write + fsync."""
        buf, k = _pack_records(items)
        if k:
            with open(self.path, "ab") as f:
                f.write(struct.pack("<III", k, len(buf), zlib.crc32(bytes(buf)) & 0xFFFFFFFF) + bytes(buf))
                f.flush()
                os.fsync(f.fileno())
            self.records += k

    def compact(self, live_items):
        """Rewrite the log with the given live (key, value) pairs as one group (atomic rename)."""
        tmp = self.path + ".tmp"
        buf, k = _pack_records(live_items)
        with open(tmp, "wb") as f:
            f.write(MAGIC)
            if k:
                f.write(struct.pack("<III", k, len(buf), zlib.crc32(bytes(buf)) & 0xFFFFFFFF))
                f.write(buf)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path)
        self.records = k
