"""Calibration-file and L1/L2 file access for the host side.

The reference opens every file with ``asdf.open(path)`` and indexes ``f["roman"][...]``
(e.g. ``utils/ipc_linearity.py:170,183,324``, ``utils/fitting.py:201,207``).  ``open_tree`` gives the
same mapping for
  * a ``dict`` (already in memory: ``{"roman": {...}}`` or the ``roman`` branch itself),
  * an ``.npz`` mirror written by ``save_npz_tree`` (keys are '/'-joined paths),
  * an ``.asdf`` file: through the ``asdf`` package when it is installed, otherwise through the small
    reader below (ASDF 1.x: YAML tree + binary blocks, uncompressed / zlib / bzip2 blocks, ndarray
    tags with ``source``/``datatype``/``byteorder``/``shape``[/``offset``/``strides``], inline arrays).
``write_asdf`` writes such files (uncompressed blocks) so that synthetic CALDIR sets and L2 products
can be exchanged with the reference tooling.
"""

import bz2
import contextlib
import hashlib
import io
import json
import os
import struct
import zlib

import numpy as np
import yaml

_BLOCK_MAGIC = b"\xd3BLK"
_YAML_END = b"\n...\n"

_DT = {
    "int8": "i1", "int16": "i2", "int32": "i4", "int64": "i8", "uint8": "u1", "uint16": "u2", "uint32": "u4",
    "uint64": "u8", "float16": "f2", "float32": "f4", "float64": "f8", "complex64": "c8", "complex128": "c16",
    "bool8": "b1",
}
_DT_INV = {np.dtype(v).name: k for k, v in _DT.items()}
_DT_INV["bool"] = "bool8"


# ------------------------------------------------------------------------------- reading
class _Loader(yaml.SafeLoader):
    pass


class _NDArrayNode(dict):
    """mapping of an ndarray tag, resolved against the block list after the YAML pass"""


def _construct_any(loader, suffix, node):
    is_nd = "core/ndarray" in suffix or "core/ndarray" in getattr(node, "tag", "")
    if isinstance(node, yaml.MappingNode):
        m = loader.construct_mapping(node, deep=True)
        return _NDArrayNode(m) if is_nd else m
    if isinstance(node, yaml.SequenceNode):
        seq = loader.construct_sequence(node, deep=True)
        return _NDArrayNode({"data": seq}) if is_nd else seq
    return loader.construct_scalar(node)


_Loader.add_multi_constructor("", _construct_any)
_Loader.add_multi_constructor("!", _construct_any)
_Loader.add_multi_constructor("tag:", _construct_any)


def _read_blocks(buf, pos):
    blocks = []
    n = len(buf)
    while True:
        pos = buf.find(_BLOCK_MAGIC, pos)
        if pos < 0 or pos + 6 > n:
            break
        (hsize,) = struct.unpack(">H", buf[pos + 4 : pos + 6])
        hdr = buf[pos + 6 : pos + 6 + hsize]
        flags, comp, alloc, used, dsize = struct.unpack(">I4sQQQ", hdr[:32])
        start = pos + 6 + hsize
        raw = bytes(buf[start : start + used])
        comp = comp.rstrip(b"\0 ")
        if comp == b"zlib":
            raw = zlib.decompress(raw)
        elif comp == b"bzp2":
            raw = bz2.decompress(raw)
        elif comp not in (b"",):
            raise NotImplementedError(f"ASDF block compression {comp!r} is not supported by the in-repo reader")
        blocks.append(raw)
        pos = start + alloc
    return blocks


def _resolve(node, blocks):
    if isinstance(node, _NDArrayNode):
        dt = node.get("datatype", "float64")
        if not isinstance(dt, str):
            raise NotImplementedError("structured ASDF datatypes are not supported by the in-repo reader")
        dtype = np.dtype(_DT[dt]).newbyteorder("<" if node.get("byteorder", "little") == "little" else ">")
        if "source" in node:
            raw = blocks[int(node["source"])]
            shape = tuple(int(s) for s in node.get("shape", [len(raw) // dtype.itemsize]))
            off = int(node.get("offset", 0))
            if "strides" in node:
                arr = np.ndarray(shape, dtype=dtype, buffer=raw, offset=off, strides=tuple(node["strides"]))
            else:
                arr = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape, dtype=np.int64)), offset=off).reshape(shape)
            return arr.astype(dtype.newbyteorder("="), copy=False)
        return np.array(node["data"], dtype=dtype.newbyteorder("="))
    if isinstance(node, dict):
        return {k: _resolve(v, blocks) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, blocks) for v in node]
    return node


def read_asdf(path):
    """Parse an ASDF file into nested dicts / lists / numpy arrays (in-repo reader)."""
    with open(path, "rb") as f:
        buf = f.read()
    if not buf.startswith(b"#ASDF"):
        raise ValueError(f"{path} is not an ASDF file")
    ystart = buf.find(b"%YAML")
    yend = buf.find(_YAML_END, ystart)
    if ystart < 0 or yend < 0:
        raise ValueError(f"{path}: no YAML tree found")
    tree = yaml.load(io.BytesIO(buf[ystart : yend + 1]), Loader=_Loader)  # noqa: S506 (SafeLoader subclass)
    blocks = _read_blocks(buf, yend + len(_YAML_END))
    return _resolve(tree, blocks)


# ------------------------------------------------------------------------------- writing
class _Dumper(yaml.SafeDumper):
    pass


class _ArrRef:
    def __init__(self, idx, arr):
        self.idx, self.arr = idx, arr


def _repr_arr(dumper, ref):
    a = ref.arr
    m = {"source": ref.idx, "datatype": _DT_INV[a.dtype.name], "byteorder": "little", "shape": [int(s) for s in a.shape]}
    return dumper.represent_mapping("!core/ndarray-1.0.0", m, flow_style=False)


_Dumper.add_representer(_ArrRef, _repr_arr)
_Dumper.add_representer(np.float64, lambda d, v: d.represent_float(float(v)))
_Dumper.add_representer(np.float32, lambda d, v: d.represent_float(float(v)))
_Dumper.add_representer(np.int64, lambda d, v: d.represent_int(int(v)))
_Dumper.add_representer(np.int32, lambda d, v: d.represent_int(int(v)))
_Dumper.add_representer(np.int16, lambda d, v: d.represent_int(int(v)))
_Dumper.add_representer(np.bool_, lambda d, v: d.represent_bool(bool(v)))


def _collect(node, blocks):
    if isinstance(node, np.ndarray):
        if node.ndim == 0:
            return node.item()
        a = np.ascontiguousarray(node)
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        blocks.append(a)
        return _ArrRef(len(blocks) - 1, a)
    if isinstance(node, dict):
        return {str(k): _collect(v, blocks) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [_collect(v, blocks) for v in node]
    return node


def write_asdf(path, tree, checksum=False):
    """Write nested dicts / lists / scalars / numpy arrays as an ASDF 1.x file with uncompressed blocks.  ``checksum``: fill
    the optional MD5 field of every block header (the standard lets it be all zero = not verified; hashing a 4096 x 4096 x 8
    exposure's L2 tree costs 0.5 s, half of a whole ``calibrateimage`` call)."""
    blocks = []
    body = _collect(tree, blocks)
    text = yaml.dump(body, Dumper=_Dumper, default_flow_style=None, sort_keys=False)
    head = (b"#ASDF 1.0.0\n#ASDF_STANDARD 1.5.0\n%YAML 1.1\n%TAG ! tag:stsci.edu:asdf/\n--- !core/asdf-1.1.0\n")
    with open(path, "wb") as f:
        f.write(head)
        f.write(text.encode("utf-8"))
        f.write(b"...\n")
        for a in blocks:
            c = np.ascontiguousarray(a)
            raw = memoryview(c.reshape(-1).view(np.uint8)) if c.size else memoryview(b"")
            digest = hashlib.md5(raw).digest() if checksum else b"\0" * 16  # noqa: S324
            hdr = struct.pack(">I4sQQQ16s", 0, b"\0\0\0\0", raw.nbytes, raw.nbytes, raw.nbytes, digest)
            f.write(_BLOCK_MAGIC + struct.pack(">H", len(hdr)) + hdr)
            f.write(raw)


# ------------------------------------------------------------------------------- npz mirrors
def save_npz_tree(path, tree):
    """Flatten a nested dict of arrays/scalars into an .npz ('/'-joined keys)."""
    flat = {}

    def walk(prefix, node):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(f"{prefix}/{k}" if prefix else str(k), v)
        elif isinstance(node, np.ndarray) or np.isscalar(node) and not isinstance(node, (str, bool)):
            flat[prefix] = np.asarray(node)
        else:  # lists (possibly ragged), strings, booleans, None: JSON text
            flat["json:" + prefix] = np.array(json.dumps(node))

    walk("", tree)
    np.savez(path, **flat)


def _load_npz_tree(path):
    tree = {}
    with np.load(path, allow_pickle=False) as f:
        for key in f.files:
            node = tree
            is_json = key.startswith("json:")
            parts = (key[5:] if is_json else key).split("/")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            v = f[key]
            node[parts[-1]] = json.loads(str(v)) if is_json else (v.item() if v.ndim == 0 else v)
    return tree


# ------------------------------------------------------------------------------- the one entry point
@contextlib.contextmanager
def open_tree(src):
    """Context manager yielding a mapping with a ``"roman"`` branch, like ``asdf.open(path)``."""
    if isinstance(src, dict):
        yield src if "roman" in src else {"roman": src}
        return
    path = os.fspath(src)
    if path.endswith(".npz"):
        t = _load_npz_tree(path)
        yield t if "roman" in t else {"roman": t}
        return
    try:
        import asdf  # noqa: PLC0415
    except ImportError:
        asdf = None
    if asdf is not None:
        with asdf.open(path) as f:
            yield f
        return
    yield read_asdf(path)


def roman_branch(src):
    """The ``roman`` branch with array leaves materialised as numpy arrays."""
    with open_tree(src) as f:
        return _materialise(f["roman"])


def _materialise(node):
    if isinstance(node, dict) or hasattr(node, "items"):
        return {k: _materialise(v) for k, v in node.items()}
    if hasattr(node, "shape") and hasattr(node, "dtype"):
        return np.asarray(node)
    return node
