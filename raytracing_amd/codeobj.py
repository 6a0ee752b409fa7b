"""Identity of the device code inside librt_hip.so: SHA-256 of the ELF section that holds the gfx950 code objects
(.hip_fatbin).  Counter files under profiles/ record it (tools/make_counters_json.py) and bench.py compares, so a
`roofline` read from counters of OTHER kernels says so (`stale: true`)."""
import hashlib
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "librt_hip.so")


def section_bytes(path, name):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError("%s is not a 64-bit ELF file" % path)
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        n, t, fl, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return n, off, size
    _, stroff, strsize = sh(shstrndx)
    names = data[stroff:stroff + strsize]
    for i in range(shnum):
        n, off, size = sh(i)
        end = names.index(b"\0", n)
        if names[n:end].decode() == name:
            return data[off:off + size]
    return None


def code_object_sha256(path=LIB):
    """hex digest of the .hip_fatbin section (None if the library is not built or has no such section)"""
    try:
        blob = section_bytes(path, ".hip_fatbin")
    except (OSError, ValueError):
        return None
    return hashlib.sha256(blob).hexdigest() if blob else None


if __name__ == "__main__":
    print(code_object_sha256())
