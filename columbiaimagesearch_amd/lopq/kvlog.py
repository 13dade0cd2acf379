"""Persistent key/value store for LOPQSearcherLMDB when the `lmdb` module is not installed.

The reference keeps its production index in LMDB (lopq/lopq/search.py:416-417): key = cell (2 x uint16,
``array('H')``) + ``bytes(id)``, value = the fine codes as bytes, ``put`` replaces an existing key (:445-470), a cell is
read back with a cursor in key order (:482-499), and one ``add_codes`` call is ONE write transaction (:459-467: ``with
env.begin(write=True)`` -- an exception inside it aborts the transaction) -- it is stored as a whole or not at all.  This
file keeps exactly those keys and values -- same bytes, same last-write-wins rule, same all-or-nothing unit -- in an
append-only log, so that an index survives a restart without LMDB:

    <lmdb_path>/index.ciskv :=  b"CISKV3\n"  group*
    group   :=  header payload
    header  :=  b"GRP3", u32le n_records, u64le payload_bytes, u32le crc32(payload), u32le flags, u32le crc32(the 24 bytes before)
    payload :=  { u32le key_len, u32le val_len, key, value } * n_records
    flags   :=  bit 0: the transaction CONTINUES in the next group (the last group of a transaction has it clear)

A transaction is one ``append`` call (= one add_codes call = one LMDB transaction).  Groups are bounded (GROUP_BYTES of
payload): a large call or a compaction is written as several groups, sizes are 64-bit, and nothing is ever held in memory
beyond one group -- the writer streams, the reader streams.  On open
  * a group that runs past the end of the file, a header cut short, or a tail of zero bytes (a file that was extended but
    never written) is a crash in the middle of a write: it and every earlier group of the SAME transaction are dropped as a
    whole and the file is cut back to the last complete transaction;
  * a header whose own checksum is wrong with real bytes behind it, or a complete group whose payload checksum or record
    framing is wrong, is corruption in the middle of the data and raises instead of silently dropping everything behind it.
``compact`` rewrites the log with one record per live key (tmp file + atomic rename).  Logs of the earlier formats
(``CISKV1``: bare records; ``CISKV2``: 32-bit groups) are read and rewritten in this one on open.
It is NOT the LMDB file format: a deployment that has LMDB files keeps using them through the `lmdb` module
(LOPQSearcherLMDB refuses to shadow an existing ``data.mdb`` with a log).
"""
import os
import struct
import zlib

MAGIC = b"CISKV3\n"
MAGIC_V2 = b"CISKV2\n"
MAGIC_V1 = b"CISKV1\n"
FILE_NAME = "index.ciskv"
GROUP_TAG = b"GRP3"
HEADER = struct.Struct("<4sIQII")  # tag, n_records, payload_bytes, crc32(payload), flags  (+ u32 crc32 of these 24 bytes)
HEADER_BYTES = HEADER.size + 4
FLAG_CONTINUES = 1
GROUP_BYTES = 64 << 20  # payload bytes per group at most (one record may exceed it on its own)


def _unpack_records(data, pos, end, n_expected=None):
    """Records of data[pos:end]; returns (records, position after the last complete record)."""
    out = []
    while pos + 8 <= end:
        kl, vl = struct.unpack_from("<II", data, pos)
        if pos + 8 + kl + vl > end:
            break
        out.append((bytes(data[pos + 8:pos + 8 + kl]), bytes(data[pos + 8 + kl:pos + 8 + kl + vl])))
        pos += 8 + kl + vl
        if n_expected is not None and len(out) == n_expected:
            break
    return out, pos


def _groups(items, group_bytes):
    """Bounded payloads of the records of `items`: yields (n_records, payload bytearray, is_last).  A full group is only
    handed out when another record follows it, so the flag is known without looking ahead."""
    buf, k = bytearray(), 0
    for key, val in items:
        if k and len(buf) + 8 + len(key) + len(val) > group_bytes:
            yield k, buf, False
            buf, k = bytearray(), 0
        buf += struct.pack("<II", len(key), len(val))
        buf += key
        buf += val
        k += 1
    if k:
        yield k, buf, True


def _write_groups(f, items, group_bytes, one_transaction):
    """Streams the records of `items` into f as bounded groups; returns the number of records written."""
    total = 0
    for k, buf, last in _groups(items, group_bytes):
        flags = FLAG_CONTINUES if (one_transaction and not last) else 0
        head = HEADER.pack(GROUP_TAG, k, len(buf), zlib.crc32(buf) & 0xFFFFFFFF, flags)
        f.write(head + struct.pack("<I", zlib.crc32(head) & 0xFFFFFFFF))
        f.write(buf)
        total += k
    return total


class KVLog(object):
    def __init__(self, path, group_bytes=GROUP_BYTES):
        self.dir = path
        os.makedirs(path, exist_ok=True)
        self.path = os.path.join(path, FILE_NAME)
        self.group_bytes = int(group_bytes)
        self.records = 0
        self.dropped_torn_bytes = 0  # what the last load() cut off the tail (a torn transaction)
        if not os.path.exists(self.path):
            with open(self.path, "wb") as f:
                f.write(MAGIC)
                f.flush()
                os.fsync(f.fileno())

    # -- earlier formats: read whole (they could not exceed 4 GiB per group anyway), rewritten in the current one ---------------
    def _load_old(self, data):
        n = len(data)
        if data[:len(MAGIC_V1)] == MAGIC_V1:  # bare records, a torn tail record is cut off
            out, good = _unpack_records(data, len(MAGIC_V1), n)
            self.dropped_torn_bytes = n - good
            return out
        pos = good = len(MAGIC_V2)
        out = []
        while pos < n:
            if pos + 12 > n:
                break
            nrec, nbytes, crc = struct.unpack_from("<III", data, pos)
            if pos + 12 + nbytes > n:
                break
            payload = data[pos + 12:pos + 12 + nbytes]
            recs, end = _unpack_records(payload, 0, nbytes, nrec) if nrec else ([], 0)
            if (zlib.crc32(payload) & 0xFFFFFFFF) != crc or len(recs) != nrec or end != nbytes:
                raise ValueError("%s: group at byte %d is damaged (checksum / framing); %d records before it are intact -- "
                                 "refusing to drop the %d bytes behind it silently" % (self.path, pos, len(out), n - pos))
            out.extend(recs)
            pos += 12 + nbytes
            good = pos
        self.dropped_torn_bytes = n - good
        return out

    def load(self):
        """(key, value) pairs in log order.  A torn last transaction is dropped as a whole (and cut off the file); damage
        in the middle of the log raises ValueError.  The file is read group by group."""
        size = os.path.getsize(self.path)
        with open(self.path, "rb") as f:
            magic = f.read(len(MAGIC))
            if magic in (MAGIC_V1, MAGIC_V2):
                out = self._load_old(magic + f.read())
                self.compact(out)
                return out
            if magic != MAGIC:
                raise ValueError("%s is not a CISKV log" % self.path)
            out = []
            pos = good = len(MAGIC)   # good: end of the last complete TRANSACTION
            n_good = 0                # records up to there
            while pos < size:
                head = f.read(HEADER_BYTES)
                if len(head) < HEADER_BYTES:
                    break  # torn header
                tag, nrec, nbytes, crc, flags = HEADER.unpack_from(head, 0)
                (hcrc,) = struct.unpack_from("<I", head, HEADER.size)
                if tag != GROUP_TAG or (zlib.crc32(head[:HEADER.size]) & 0xFFFFFFFF) != hcrc:
                    # a tail of zero bytes = a file extended by a write that never landed: torn.  Anything else: damage.
                    rest_zero = not any(head)
                    while rest_zero:
                        chunk = f.read(1 << 20)
                        if not chunk:
                            break
                        rest_zero = not any(chunk)
                    if rest_zero:
                        break
                    raise ValueError("%s: group header at byte %d is damaged; %d records before it are intact -- refusing to drop "
                                     "the %d bytes behind it silently" % (self.path, pos, n_good, size - pos))
                if pos + HEADER_BYTES + nbytes > size:
                    break  # torn group: the write of this add_codes call never completed -> its transaction is dropped
                payload = f.read(nbytes)
                recs, end = _unpack_records(payload, 0, nbytes, nrec) if nrec else ([], 0)
                if (zlib.crc32(payload) & 0xFFFFFFFF) != crc or len(recs) != nrec or end != nbytes:
                    raise ValueError("%s: group at byte %d is damaged (checksum / framing); %d records before it are intact -- "
                                     "refusing to drop the %d bytes behind it silently" % (self.path, pos, n_good, size - pos))
                out.extend(recs)
                pos += HEADER_BYTES + nbytes
                if not (flags & FLAG_CONTINUES):
                    good, n_good = pos, len(out)
        del out[n_good:]  # groups of a transaction whose last group never arrived
        self.dropped_torn_bytes = size - good
        if good != size:
            with open(self.path, "r+b") as f:
                f.truncate(good)
        self.records = len(out)
        return out

    def append(self, items):
        """items: iterable of (key bytes, value bytes) of ONE add_codes call: one transaction (bounded groups, the last one
        closes it), one fsync.  Raises before anything is written when an item is not a pair of byte strings."""
        with open(self.path, "ab") as f:
            start = f.tell()
            try:
                k = _write_groups(f, items, self.group_bytes, one_transaction=True)
                f.flush()
                os.fsync(f.fileno())
            except BaseException:
                f.flush()
                f.truncate(start)  # nothing of a failed call stays behind
                raise
        self.records += k

    def compact(self, live_items):
        """Rewrite the log with the given live (key, value) pairs, streamed as bounded groups (atomic rename)."""
        tmp = self.path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(MAGIC)
            k = _write_groups(f, live_items, self.group_bytes, one_transaction=False)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path)
        self.records = k
