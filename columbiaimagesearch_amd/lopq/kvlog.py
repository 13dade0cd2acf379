"""Persistent key/value store for LOPQSearcherLMDB when the `lmdb` module is not installed.

The reference keeps its production index in LMDB (lopq/lopq/search.py:416-417): key = cell (2 x uint16,
``array('H')``) + ``bytes(id)``, value = the fine codes as bytes, ``put`` replaces an existing key (:445-470), a cell is
read back with a cursor in key order (:482-499).  This file keeps exactly those keys and values -- same bytes, same
last-write-wins rule -- in an append-only log, so that an index survives a restart without LMDB:

    <lmdb_path>/index.ciskv :=  b"CISKV1\\n"  { u32le key_len, u32le val_len, key, value }*

A record is complete or ignored: a torn tail (crash in the middle of an append) is cut off on open.  ``load`` replays
the log (later records replace earlier ones), ``append`` adds the records of one add_codes call and fsyncs (the
reference calls env.sync() after every add_codes, :468).  ``compact`` rewrites the log with one record per live key.
It is NOT the LMDB file format: a deployment that has LMDB files keeps using them through the `lmdb` module.
"""
import os
import struct

MAGIC = b"CISKV1\n"
FILE_NAME = "index.ciskv"


class KVLog(object):
    def __init__(self, path):
        self.dir = path
        os.makedirs(path, exist_ok=True)
        self.path = os.path.join(path, FILE_NAME)
        self.records = 0
        if not os.path.exists(self.path):
            with open(self.path, "wb") as f:
                f.write(MAGIC)
                f.flush()
                os.fsync(f.fileno())

    def load(self):
        """Yield (key, value) in log order; truncates a torn tail."""
        with open(self.path, "rb") as f:
            data = f.read()
        if data[:len(MAGIC)] != MAGIC:
            raise ValueError("%s is not a CISKV1 log" % self.path)
        pos, good, n = len(MAGIC), len(MAGIC), len(data)
        out = []
        while pos + 8 <= n:
            kl, vl = struct.unpack_from("<II", data, pos)
            if pos + 8 + kl + vl > n:
                break
            out.append((data[pos + 8:pos + 8 + kl], data[pos + 8 + kl:pos + 8 + kl + vl]))
            pos += 8 + kl + vl
            good = pos
        if good != n:
            with open(self.path, "r+b") as f:
                f.truncate(good)
        self.records = len(out)
        return out

    def append(self, items):
        """items: iterable of (key bytes, value bytes); one write + fsync for the whole call."""
        buf = bytearray()
        k = 0
        for key, val in items:
            buf += struct.pack("<II", len(key), len(val))
            buf += key
            buf += val
            k += 1
        if k:
            with open(self.path, "ab") as f:
                f.write(buf)
                f.flush()
                os.fsync(f.fileno())
            self.records += k

    def compact(self, live_items):
        """Rewrite the log with the given live (key, value) pairs (atomic rename)."""
        tmp = self.path + ".tmp"
        k = 0
        with open(tmp, "wb") as f:
            f.write(MAGIC)
            for key, val in live_items:
                f.write(struct.pack("<II", len(key), len(val)))
                f.write(key)
                f.write(val)
                k += 1
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path)
        self.records = k
