"""Identity of the device code inside librt_hip.so: SHA-256 over the instructions and kernel descriptors of the gfx950
code objects in its .hip_fatbin section.  Counter files under profiles/ record it (tools/make_counters_json.py) and bench.py compares, so a
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


def _elf_sections(data):
    """{name: bytes} of a 64-bit little-endian ELF image held in memory"""
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        n, t, fl, addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return n, t, off, size
    _, _, stroff, strsize = sh(shstrndx)
    names = data[stroff:stroff + strsize]
    out = {}
    for i in range(shnum):
        n, t, off, size = sh(i)
        name = names[n:names.index(b"\0", n)].decode()
        out[name] = b"" if t == 8 else data[off:off + size]            # SHT_NOBITS has no bytes in the file
    return out


def device_code_objects(path=LIB):
    """[(target triple, ELF image)] of every clang offload bundle in .hip_fatbin (one bundle per translation unit with device code)"""
    blob = section_bytes(path, ".hip_fatbin")
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    start = blob.find(magic) if blob else -1
    while start >= 0:
        n, = struct.unpack_from("<Q", blob, start + len(magic))
        pos = start + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if size and blob[start + off:start + off + 4] == b"\x7fELF":
                out.append((triple, blob[start + off:start + off + size]))
        start = blob.find(magic, pos)
    return out


HOT_PATH_KERNEL = b"k_trace_w4"      # the code object that holds the hot path (rt_hip.hip); device_fold.hip's is another one


def code_object_sha256(path=LIB):
    """hex digest of the device CODE of the library: the .text (instructions) and .rodata (kernel descriptors) sections of
    the gfx code object(s) in .hip_fatbin that hold the hot path's kernels.  Symbol tables are left out on purpose: clang names a per-translation-unit symbol
    (__hip_cuid_...) after the source PATH, so the same source built in another directory differs there and nowhere else.
    None if the library is not built or holds no code object."""
    try:
        objs = device_code_objects(path)
    except (OSError, ValueError, struct.error):
        return None
    if not objs:
        return None
    # (round 6: the library has two code objects -- the hot path's and the tree builder's; the counters and the bench line speak for the first)
    hot = [(t, img) for t, img in objs if HOT_PATH_KERNEL in img]
    objs = hot or objs
    h = hashlib.sha256()
    for triple, image in objs:
        sec = _elf_sections(image)
        h.update(triple.encode())
        for name in (".text", ".rodata"):
            h.update(name.encode())
            h.update(sec.get(name, b""))
    return h.hexdigest()


if __name__ == "__main__":
    print(code_object_sha256())
