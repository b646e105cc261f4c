"""Minimal Ogg Opus reading / writing helpers (RFC 3533, RFC 7845) for building TEST streams:
split a file into packets, and mux packets of several elementary Opus streams into one multistream
(channel mapping family 1) file using self-delimiting framing (RFC 6716 appendix B)."""
import struct


def _crc_table():
    t = []
    for i in range(256):
        r = i << 24
        for _ in range(8):
            r = ((r << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if r & 0x80000000 else (r << 1) & 0xFFFFFFFF
        t.append(r)
    return t


_T = _crc_table()


def ogg_crc(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFFFFFF) ^ _T[((c >> 24) & 0xFF) ^ b]
    return c


def read_packets(raw):
    """-> (packets, granule of the page each packet ENDS on, last granule)"""
    pos, pend, packets, grans = 0, b"", [], []
    last = -1
    while pos + 27 <= len(raw) and raw[pos:pos + 4] == b"OggS":
        gran, = struct.unpack("<q", raw[pos + 6:pos + 14])
        nseg = raw[pos + 26]
        lac = raw[pos + 27:pos + 27 + nseg]
        body = pos + 27 + nseg
        off = 0
        for s in lac:
            pend += raw[body + off:body + off + s]
            off += s
            if s < 255:
                packets.append(pend)
                grans.append(gran)
                pend = b""
        if gran >= 0:
            last = gran
        pos = body + off
    return packets, grans, last


def page(serial, seq, granule, packet, flags):
    lac = [255] * (len(packet) // 255) + [len(packet) % 255]
    assert len(lac) <= 255
    hdr = b"OggS" + bytes([0, flags]) + struct.pack("<qIII", granule, serial, seq, 0) + bytes([len(lac)]) + bytes(lac)
    crc = ogg_crc(hdr + packet)
    return hdr[:22] + struct.pack("<I", crc) + hdr[26:] + packet


def self_delimit(pkt):
    """Re-frame a code-0 Opus packet (one frame) with an explicit frame length."""
    assert (pkt[0] & 3) == 0, "only single-frame packets are re-framed by this helper"
    n = len(pkt) - 1
    ln = bytes([n]) if n < 252 else bytes([252 + (n & 3), (n - (252 + (n & 3))) >> 2])
    return pkt[:1] + ln + pkt[1:]


def mux_family1(streams, coupled, mapping, preskip, granules, serial=0x4E595131):
    """streams: list of packet lists (equal length); coupled: number of leading stereo streams."""
    nch = len(mapping)
    head = (b"OpusHead" + bytes([1, nch]) + struct.pack("<HIh", preskip, 48000, 0) + bytes([1, len(streams), coupled])
            + bytes(mapping))
    tags = b"OpusTags" + struct.pack("<I", 8) + b"nyq-test" + struct.pack("<I", 0)
    out = [page(serial, 0, 0, head, 2), page(serial, 1, 0, tags, 0)]
    n = len(streams[0])
    for i in range(n):
        pkt = b"".join(self_delimit(s[i]) for s in streams[:-1]) + streams[-1][i]
        out.append(page(serial, 2 + i, granules[i], pkt, 4 if i == n - 1 else 0))
    return b"".join(out)


def mux_family0(packets, channels, preskip, frame_samples, total_samples, serial=0x4E595130, per_page=8):
    """Single mono/stereo Opus stream (channel mapping family 0).  `total_samples` = input samples per
    channel: the last page's granule position trims the encoder's padding (RFC 7845 section 4.4)."""
    head = b"OpusHead" + bytes([1, channels]) + struct.pack("<HIh", preskip, 48000, 0) + bytes([0])
    return mux_packets(head, packets, preskip, frame_samples, total_samples, serial, per_page)


def mux_packets(head, packets, preskip, frame_samples, total_samples, serial=0x4E595130, per_page=8):
    """Ogg pages around ready-made Opus packets (any mapping family: `head` is the OpusHead packet)."""
    tags = b"OpusTags" + struct.pack("<I", 8) + b"nyq-test" + struct.pack("<I", 0)
    out = [page(serial, 0, 0, head, 2), page(serial, 1, 0, tags, 0)]
    seq, done, i, n = 2, 0, 0, len(packets)
    while i < n:
        group = packets[i:i + per_page]
        i += len(group)
        done += len(group) * frame_samples
        last = i >= n
        gran = min(done, preskip + total_samples) if last else done
        lac, body = [], b""
        for pk in group:
            lac += [255] * (len(pk) // 255) + [len(pk) % 255]
            body += pk
        assert len(lac) <= 255
        hdr = b"OggS" + bytes([0, 4 if last else 0]) + struct.pack("<qIII", gran, serial, seq, 0) + bytes([len(lac)]) + bytes(lac)
        crc = ogg_crc(hdr + body)
        out.append(hdr[:22] + struct.pack("<I", crc) + hdr[26:] + body)
        seq += 1
    return b"".join(out)


def mux_packets_sized(head, packets, sizes, preskip, total_samples, serial=0x4E595132, per_page=8):
    """like mux_packets, for packets of different durations (`sizes[i]` samples at 48 kHz)"""
    tags = b"OpusTags" + struct.pack("<I", 8) + b"nyq-test" + struct.pack("<I", 0)
    out = [page(serial, 0, 0, head, 2), page(serial, 1, 0, tags, 0)]
    seq, done, i, n = 2, 0, 0, len(packets)
    while i < n:
        group = packets[i:i + per_page]
        done += sum(sizes[i:i + len(group)])
        i += len(group)
        last = i >= n
        gran = min(done, preskip + total_samples) if last else done
        lac, body = [], b""
        for pk in group:
            lac += [255] * (len(pk) // 255) + [len(pk) % 255]
            body += pk
        assert len(lac) <= 255
        hdr = b"OggS" + bytes([0, 4 if last else 0]) + struct.pack("<qIII", gran, serial, seq, 0) + bytes([len(lac)]) + bytes(lac)
        crc = ogg_crc(hdr + body)
        out.append(hdr[:22] + struct.pack("<I", crc) + hdr[26:] + body)
        seq += 1
    return b"".join(out)
