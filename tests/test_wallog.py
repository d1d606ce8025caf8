"""The WAL backer file image (`smr_wallog_*`, summerset_amd/csrc/wire.hip, host only) against the reference's OWN unit tests
of StorageHubLoggerTask's file operations (src/server/storage.rs:560-790: write_entries, append_entries, read_entries,
truncate_log, discard_log), restated call by call.  TestEntry(String) under bincode's standard config is a varint length
followed by the UTF-8 bytes."""
import pytest

from summerset_amd import wire

ENTRY = bytes([len(b"test-entry-dummy-string")]) + b"test-entry-dummy-string"      # bincode of TestEntry("test-entry-dummy-string")


def test_write_entries():                                   # storage.rs:586-630
    f = wire.WalLog()
    ok, now = f.write_entry(0, ENTRY, 0)
    assert ok and now == 8 + len(ENTRY)
    ok, now = f.write_entry(now, ENTRY, now)                # at the end
    assert ok and now == 2 * (8 + len(ENTRY))
    ok, now2 = f.write_entry(now, ENTRY, 0)                 # over the first entry: the size stays
    assert ok and now2 == now
    ok, now3 = f.write_entry(now2, ENTRY, now2 + 10)        # a hole: refused (:289-297)
    assert not ok and now3 == now2
    assert f.bytes() == ((len(ENTRY)).to_bytes(8, "big") + ENTRY) * 2


def test_append_entries():                                  # storage.rs:632-655
    f = wire.WalLog()
    mid = f.append_entry(0, ENTRY)
    assert mid >= len(ENTRY)
    end = f.append_entry(mid, ENTRY)
    assert end - mid >= len(ENTRY)
    assert (mid, end) == (8 + len(ENTRY), 2 * (8 + len(ENTRY))) and len(f) == end


def test_read_entries():                                    # storage.rs:657-710
    f = wire.WalLog()
    mid = f.append_entry(0, ENTRY)
    end = f.append_entry(mid, ENTRY)
    assert f.read_entry(end, mid) == (ENTRY, end)
    assert f.read_entry(end, 0) == (ENTRY, mid)
    assert f.read_entry(end, mid + 10) == (None, mid + 10)  # lands inside an entry: the "length" there runs past the file
    assert f.read_entry(mid, mid - 4) == (None, mid - 4)    # header would cross the file bound


def test_truncate_log():                                    # storage.rs:712-760 (the reference's fn is named truncate_log too)
    f = wire.WalLog()
    mid = f.append_entry(0, ENTRY)
    end = f.append_entry(mid, ENTRY)
    assert f.truncate_log(end, mid) == (True, mid)
    assert f.truncate_log(mid, end) == (False, mid)
    assert f.truncate_log(mid, 0) == (True, 0)
    assert len(f) == 0


def test_discard_log():                                     # storage.rs:762-840
    f = wire.WalLog()
    mid1 = f.append_entry(0, ENTRY)
    mid2 = f.append_entry(mid1, ENTRY)
    end = f.append_entry(mid2, ENTRY)
    tail = end - mid2
    assert f.discard_log(end, mid2, mid1) == (True, 2 * tail)
    assert f.discard_log(2 * tail, mid1, end) == (False, 2 * tail)      # keep >= offset
    assert f.discard_log(2 * tail, mid1, 0) == (True, tail)
    assert f.discard_log(tail, end, 0) == (False, tail)                 # offset beyond the file
    assert f.discard_log(tail, tail, 0) == (True, 0)
    assert len(f) == 0


def test_the_engines_own_wal_entries_go_through_the_log():
    """PrepareBal / AcceptData / CommitSlot entries as the encoders make them, appended and read back (recovery's loop:
    read_entry from offset 0 until None, multipaxos/recovery.rs:119-150)"""
    f = wire.WalLog()
    entries = [wire.frame_payload(wire.wal_prepare_bal(3, 0x102)), wire.frame_payload(wire.wal_accept_data(3, 0x102, b"\x00")),
               wire.frame_payload(wire.wal_commit_slot(3))]
    size = 0
    for e in entries:
        size = f.append_entry(size, e)
    got, off = [], 0
    while True:
        e, off2 = f.read_entry(size, off)
        if e is None:
            break
        got.append(e); off = off2
    assert got == entries and off == size
    # the appended frames are byte for byte what the encoders emit with their header
    assert f.bytes() == wire.wal_prepare_bal(3, 0x102) + wire.wal_accept_data(3, 0x102, b"\x00") + wire.wal_commit_slot(3)


def test_argument_errors():
    from summerset_amd import SummersetError
    f = wire.WalLog()
    f.append_entry(0, ENTRY)
    with pytest.raises(SummersetError):
        f.read_entry(1000, 500)                              # a file_size the image does not have
