"""TEST INFRASTRUCTURE (never imported by the product): a small SPIR-V interpreter that executes the
reference's own compiled shaders -- shaders/raygen.rgen.spv, closesthit.rchit.spv, miss.rmiss.spv, the binaries
main.cpp:541-543 loads into the ray-tracing pipeline -- one launch invocation at a time on the CPU.

Purpose: pin `oracle/pt_oracle.c`.  The reference cannot be built here (Vulkan SDK, RT driver, GLFW), but its
shader *binaries* are committed, and they are the whole radiance loop.  Running them through this VM and comparing
with the oracle's per-pixel results bit for bit checks every shader-level decision of the restatement (seed
derivation, order of the rand() calls, camera arithmetic, loop structure, hit shading, miss, throughput update,
accumulation) against the reference itself rather than against a reading of its source.

What the VM has to supply, because Vulkan leaves it to the implementation, comes from the project's canonical
arithmetic through the `Driver` object:
  * OpTraceRayKHR (traceRayEXT, raygen.rgen:63-75): closest hit -> (primitive id, barycentrics) or miss;
  * GLSL.std.450 Sin/Cos/Sqrt/Normalize/Cross/FAbs and OpDot: evaluation order and rounding;
  * the storage image (format conversion on imageLoad/imageStore).
Float arithmetic is IEEE binary32, one rounding per SPIR-V instruction (numpy float32 scalars), no contraction:
SPIR-V has no fused multiply-add unless the shader asks for one, and these shaders do not.

The module reads `.spv` files given by path; nothing of the reference is stored in this repository.  It runs only
where /root/reference exists (this container): `tests/golden/make_spirv_goldens.py` turns its outputs into the
committed fixture `tests/golden/spirv_pixels.npz`.
"""
import struct

import numpy as np

F32 = np.float32
M32 = 0xFFFFFFFF

# storage classes / builtins / decorations (SPIR-V 1.6 unified spec + SPV_KHR_ray_tracing)
SC_UNIFORM_CONSTANT, SC_INPUT, SC_UNIFORM, SC_PRIVATE, SC_FUNCTION, SC_PUSH_CONSTANT, SC_STORAGE_BUFFER = 0, 1, 2, 6, 7, 9, 12
SC_RAY_PAYLOAD, SC_HIT_ATTRIBUTE, SC_INCOMING_RAY_PAYLOAD = 5338, 5339, 5342
DEC_BUILTIN, DEC_LOCATION, DEC_BINDING = 11, 30, 33
BI_PRIMITIVE_ID, BI_LAUNCH_ID, BI_LAUNCH_SIZE = 7, 5319, 5320
GLSL_FABS, GLSL_SIN, GLSL_COS, GLSL_SQRT, GLSL_CROSS, GLSL_NORMALIZE = 4, 13, 14, 31, 68, 69


class Cell:
    """one variable: a value tree (nested lists of scalars)"""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


class Ptr:
    __slots__ = ("cell", "path")

    def __init__(self, cell, path=()):
        self.cell = cell
        self.path = path

    def load(self):
        v = self.cell.v
        for i in self.path:
            v = v[i]
        return _copy(v)

    def store(self, val):
        val = _copy(val)
        if not self.path:
            self.cell.v = val
            return
        v = self.cell.v
        for i in self.path[:-1]:
            v = v[i]
        v[self.path[-1]] = val


def _copy(v):
    return [_copy(x) for x in v] if isinstance(v, list) else v


def _ew1(f, a):
    return [f(x) for x in a] if isinstance(a, list) else f(a)


def _ew2(f, a, b):
    if isinstance(a, list):
        return [f(x, y) for x, y in zip(a, b)]
    return f(a, b)


def _signed(u):
    return u - (1 << 32) if u & 0x80000000 else u


class Function:
    def __init__(self, fid, rtype):
        self.id, self.rtype, self.params, self.blocks, self.entry = fid, rtype, [], {}, None


class Module:
    """parsed SPIR-V module (only what the three reference shaders use; anything else raises)"""

    def __init__(self, path, int_const_override=None):
        """int_const_override: {value: new value} applied to scalar OpConstants of integer type -- the one use is
        specialising `int maxSamples = 32` (raygen.rgen:43, a single OpStore of that constant) to another sample
        count so a launch can be compared at 1 spp; the instruction stream is untouched."""
        blob = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(blob) // 4), blob)
        if w[0] != 0x07230203:
            raise ValueError("not a SPIR-V module: " + path)
        self.version, self.bound = w[1], w[3]
        self.types, self.names, self.decor, self.globals_, self.functions = {}, {}, {}, {}, {}
        self.const = [None] * self.bound
        self.entry = None
        self.ext_glsl = None
        cur, block = None, None
        i = 5
        while i < len(w):
            op, n = w[i] & 0xFFFF, w[i] >> 16
            o = w[i + 1:i + n]
            i += n
            if op in (3, 4, 6, 7, 8, 10, 14, 16, 17, 72, 330):  # source/debug, extension, memory model, capability, member decor
                continue
            if op == 5:
                self.names[o[0]] = _string(o[1:])
            elif op == 11:
                if _string(o[1:]) == "GLSL.std.450":
                    self.ext_glsl = o[0]
            elif op == 15:
                self.entry = o[1]
            elif op == 71:
                self.decor.setdefault(o[0], {})[o[1]] = o[2:]
            elif op == 19: self.types[o[0]] = ("void",)
            elif op == 20: self.types[o[0]] = ("bool",)
            elif op == 21: self.types[o[0]] = ("int", o[1], o[2])
            elif op == 22: self.types[o[0]] = ("float", o[1])
            elif op == 23: self.types[o[0]] = ("vec", o[1], o[2])
            elif op == 25: self.types[o[0]] = ("image",)
            elif op == 29: self.types[o[0]] = ("rtarray", o[1])
            elif op == 30: self.types[o[0]] = ("struct", o[1:])
            elif op == 32: self.types[o[0]] = ("ptr", o[1], o[2])
            elif op == 33: self.types[o[0]] = ("func", o[1], o[2:])
            elif op == 5341: self.types[o[0]] = ("accel",)
            elif op == 41: self.const[o[1]] = True
            elif op == 42: self.const[o[1]] = False
            elif op == 43:
                t = self.types[o[0]]
                self.const[o[1]] = F32(struct.unpack("<f", struct.pack("<I", o[2]))[0]) if t[0] == "float" else o[2]
                if t[0] == "int" and int_const_override and o[2] in int_const_override:
                    self.const[o[1]] = int_const_override[o[2]] & M32
            elif op == 44:
                self.const[o[1]] = [self.const[c] for c in o[2:]]
            elif op == 54:
                cur = Function(o[1], o[0])
                self.functions[o[1]] = cur
            elif op == 55:
                cur.params.append(o[1])
            elif op == 56:
                cur = None
            elif op == 59 and cur is None:
                self.globals_[o[1]] = (o[0], o[2], o[3] if len(o) > 3 else None)
            elif op == 248:
                block = []
                cur.blocks[o[0]] = block
                if cur.entry is None:
                    cur.entry = o[0]
            elif cur is not None:
                block.append((op, o))
            else:
                raise NotImplementedError("SPIR-V opcode %d at module scope" % op)

    def zero(self, tid):
        t = self.types[tid]
        k = t[0]
        if k == "float": return F32(0)
        if k == "int": return 0
        if k == "bool": return False
        if k == "vec": return [self.zero(t[1]) for _ in range(t[2])]
        if k == "struct": return [self.zero(m) for m in t[1]]
        raise NotImplementedError("zero of " + k)

    def global_by(self, storage=None, builtin=None, binding=None):
        for gid, (tid, sc, _) in self.globals_.items():
            d = self.decor.get(gid, {})
            if storage is not None and sc != storage: continue
            if builtin is not None and d.get(DEC_BUILTIN, (None,))[0] != builtin: continue
            if binding is not None and d.get(DEC_BINDING, (None,))[0] != binding: continue
            return gid
        return None


def _string(ws):
    b = b"".join(struct.pack("<I", x) for x in ws)
    return b.split(b"\0", 1)[0].decode()


class Driver:
    """what the Vulkan implementation decides: override to choose.  Defaults = DESIGN.md section 3."""

    def __init__(self, sincos):
        self._sincos = sincos  # callable(float32) -> (sin, cos), the canonical one from the oracle library

    def sin(self, a): return F32(self._sincos(a)[0])
    def cos(self, a): return F32(self._sincos(a)[1])
    def sqrt(self, a): return np.sqrt(a)  # correctly rounded binary32
    def fabs(self, a): return np.abs(a)
    def dot(self, a, b): return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
    def cross(self, a, b): return [a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]]
    def normalize(self, v): l = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); return [v[0] / l, v[1] / l, v[2] / l]
    def trace(self, origin, tmin, direction, tmax):
        """-> None (miss) or (primitive id, attribs.x, attribs.y)"""
        raise NotImplementedError
    def image_load(self, x, y): raise NotImplementedError
    def image_store(self, x, y, texel): raise NotImplementedError


class Pipeline:
    """the ray-tracing pipeline of main.cpp:540-608: one raygen, one miss, one closest-hit group, and the
    descriptor set of main.cpp:610-641 (bindings 2/3/4 = vertices / indices / faces as flat arrays)."""

    def __init__(self, rgen, rchit, rmiss, vertices, indices, faces, driver, contract=False, rgen_int_const_override=None):
        """contract: evaluate every OpFAdd / OpFSub one of whose operands is the result of an OpFMul /
        OpVectorTimesScalar of the same function as ONE fused multiply-add (product exact, one rounding) -- what a
        Vulkan compiler may do to these shaders, none of whose results is decorated NoContraction."""
        self.rgen, self.rchit, self.rmiss, self.drv = Module(rgen, rgen_int_const_override), Module(rchit), Module(rmiss), driver
        self.contract = contract
        self.buffers = {2: [[F32(x) for x in vertices]], 3: [[int(x) for x in indices]], 4: [[F32(x) for x in faces]]}
        self.n_traces = 0
        self.n_instructions = 0

    # -- one raygen invocation (gl_LaunchIDEXT = (x, y, 0)) -------------------------------
    def launch(self, x, y, width, height, frame):
        m = self.rgen
        vals = list(m.const)
        payload = None
        for gid, (tid, sc, init) in m.globals_.items():
            pointee = m.types[tid][2]
            d = m.decor.get(gid, {})
            if sc == SC_INPUT:
                bi = d[DEC_BUILTIN][0]
                v = {BI_LAUNCH_ID: [x, y, 0], BI_LAUNCH_SIZE: [width, height, 1]}[bi]
            elif sc == SC_PUSH_CONSTANT:
                v = [frame & M32]
            elif sc == SC_UNIFORM_CONSTANT:
                v = m.types[pointee][0]  # "accel" / "image": opaque handles
            elif sc == SC_RAY_PAYLOAD:
                v = m.zero(pointee)
            elif sc == SC_PRIVATE:
                v = _copy(m.const[init]) if init is not None else m.zero(pointee)
            else:
                raise NotImplementedError("raygen global in storage class %d" % sc)
            vals[gid] = Ptr(Cell(v))
        self._call(m, m.functions[m.entry], [], vals)

    def _run_hit_or_miss(self, m, payload_cell, prim=None, attribs=None):
        vals = list(m.const)
        for gid, (tid, sc, init) in m.globals_.items():
            pointee = m.types[tid][2]
            d = m.decor.get(gid, {})
            if sc == SC_INCOMING_RAY_PAYLOAD:
                vals[gid] = Ptr(payload_cell)
                continue
            if sc == SC_INPUT:
                assert d[DEC_BUILTIN][0] == BI_PRIMITIVE_ID
                v = prim
            elif sc == SC_HIT_ATTRIBUTE:
                v = [F32(attribs[0]), F32(attribs[1])]
            elif sc in (SC_STORAGE_BUFFER, SC_UNIFORM):
                v = self.buffers[d[DEC_BINDING][0]]
            elif sc == SC_PRIVATE:
                v = _copy(m.const[init]) if init is not None else m.zero(pointee)
            else:
                raise NotImplementedError("hit/miss global in storage class %d" % sc)
            vals[gid] = Ptr(Cell(v))
        self._call(m, m.functions[m.entry], [], vals)

    # -- the interpreter proper --------------------------------------------------------------
    def _call(self, m, fn, args, gvals):
        vals = list(gvals)
        for p, a in zip(fn.params, args):
            vals[p] = a
        drv = self.drv
        label = fn.entry
        count = 0
        prod = {}  # contract: result id of a multiply -> its operands

        def fused(pid, c, sign_p, sign_c):  # sign_p * (a * b) + sign_c * c with one rounding
            a, b = prod[pid]
            f = lambda x, y, z: F32(sign_p * (np.float64(x) * np.float64(y)) + sign_c * np.float64(z))  # noqa: E731
            if isinstance(a, list):
                bb = b if isinstance(b, list) else [b] * len(a)
                return [f(x, y, z) for x, y, z in zip(a, bb, c)]
            return f(a, b, c)

        while True:
            for op, o in fn.blocks[label]:
                count += 1
                if op == 61:  # OpLoad
                    vals[o[1]] = vals[o[2]].load()
                elif op == 62:  # OpStore
                    vals[o[0]].store(vals[o[1]])
                elif op == 65:  # OpAccessChain
                    b = vals[o[2]]
                    vals[o[1]] = Ptr(b.cell, b.path + tuple(vals[k] for k in o[3:]))
                elif op == 59:  # OpVariable (Function storage)
                    pointee = m.types[o[0]][2]
                    vals[o[1]] = Ptr(Cell(_copy(vals[o[3]]) if len(o) > 3 else m.zero(pointee)))
                elif op == 133:  # OpFMul
                    vals[o[1]] = _ew2(lambda a, b: a * b, vals[o[2]], vals[o[3]])
                    if self.contract: prod[o[1]] = (vals[o[2]], vals[o[3]])
                elif op == 129:  # OpFAdd
                    if self.contract and o[2] in prod: vals[o[1]] = fused(o[2], vals[o[3]], 1.0, 1.0)
                    elif self.contract and o[3] in prod: vals[o[1]] = fused(o[3], vals[o[2]], 1.0, 1.0)
                    else: vals[o[1]] = _ew2(lambda a, b: a + b, vals[o[2]], vals[o[3]])
                elif op == 131:  # OpFSub
                    if self.contract and o[2] in prod: vals[o[1]] = fused(o[2], vals[o[3]], 1.0, -1.0)
                    elif self.contract and o[3] in prod: vals[o[1]] = fused(o[3], vals[o[2]], -1.0, 1.0)
                    else: vals[o[1]] = _ew2(lambda a, b: a - b, vals[o[2]], vals[o[3]])
                elif op == 136: vals[o[1]] = _ew2(lambda a, b: a / b, vals[o[2]], vals[o[3]])  # OpFDiv
                elif op == 127: vals[o[1]] = _ew1(lambda a: -a, vals[o[2]])  # OpFNegate
                elif op == 142:  # OpVectorTimesScalar
                    s = vals[o[3]]
                    vals[o[1]] = [a * s for a in vals[o[2]]]
                    if self.contract: prod[o[1]] = (vals[o[2]], s)
                elif op == 148: vals[o[1]] = drv.dot(vals[o[2]], vals[o[3]])  # OpDot
                elif op == 128: vals[o[1]] = _ew2(lambda a, b: (a + b) & M32, vals[o[2]], vals[o[3]])  # OpIAdd
                elif op == 132: vals[o[1]] = _ew2(lambda a, b: (a * b) & M32, vals[o[2]], vals[o[3]])  # OpIMul
                elif op == 194: vals[o[1]] = _ew2(lambda a, b: a >> (b & 31), vals[o[2]], vals[o[3]])  # OpShiftRightLogical
                elif op == 198: vals[o[1]] = _ew2(lambda a, b: a ^ b, vals[o[2]], vals[o[3]])  # OpBitwiseXor
                elif op == 176: vals[o[1]] = _ew2(lambda a, b: a < b, vals[o[2]], vals[o[3]])  # OpULessThan
                elif op == 186: vals[o[1]] = _ew2(lambda a, b: bool(a > b), vals[o[2]], vals[o[3]])  # OpFOrdGreaterThan
                elif op == 112: vals[o[1]] = _ew1(lambda a: F32(a), vals[o[2]])  # OpConvertUToF: exact int -> one RNE rounding
                elif op == 111: vals[o[1]] = _ew1(lambda a: F32(_signed(a)), vals[o[2]])  # OpConvertSToF
                elif op == 124:  # OpBitcast: only int <-> uint here (same 32-bit pattern)
                    src = vals[o[2]]
                    probe = src[0] if isinstance(src, list) else src
                    if isinstance(probe, (float, np.floating)):
                        raise NotImplementedError("float bitcast")
                    vals[o[1]] = _copy(src)
                elif op == 80:  # OpCompositeConstruct
                    t = m.types[o[0]]
                    if t[0] == "vec":
                        out = []
                        for k in o[2:]:
                            v = vals[k]
                            out.extend(v) if isinstance(v, list) else out.append(v)
                        assert len(out) == t[2]
                        vals[o[1]] = out
                    else:
                        vals[o[1]] = [_copy(vals[k]) for k in o[2:]]
                elif op == 81:  # OpCompositeExtract
                    v = vals[o[2]]
                    for k in o[3:]:
                        v = v[k]
                    vals[o[1]] = _copy(v)
                elif op == 79:  # OpVectorShuffle
                    both = vals[o[2]] + vals[o[3]]
                    vals[o[1]] = [both[k] for k in o[4:]]
                elif op == 12:  # OpExtInst
                    if o[2] != m.ext_glsl:
                        raise NotImplementedError("extended instruction set")
                    e, a = o[3], [vals[k] for k in o[4:]]
                    if e == GLSL_FABS: r = _ew1(drv.fabs, a[0])
                    elif e == GLSL_SQRT: r = _ew1(drv.sqrt, a[0])
                    elif e == GLSL_SIN: r = _ew1(drv.sin, a[0])
                    elif e == GLSL_COS: r = _ew1(drv.cos, a[0])
                    elif e == GLSL_CROSS: r = drv.cross(a[0], a[1])
                    elif e == GLSL_NORMALIZE: r = drv.normalize(a[0])
                    else: raise NotImplementedError("GLSL.std.450 instruction %d" % e)
                    vals[o[1]] = r
                elif op == 57:  # OpFunctionCall
                    self.n_instructions += count
                    count = 0
                    vals[o[1]] = self._call(m, m.functions[o[2]], [vals[k] for k in o[3:]], gvals)
                elif op == 4445:  # OpTraceRayKHR: accel flags cull sbtOffset sbtStride missIndex origin tmin dir tmax payload
                    self.n_traces += 1
                    hit = drv.trace(vals[o[6]], vals[o[7]], vals[o[8]], vals[o[9]])
                    cell = vals[o[10]].cell
                    if hit is None:
                        self._run_hit_or_miss(self.rmiss, cell)
                    else:
                        self._run_hit_or_miss(self.rchit, cell, prim=int(hit[0]) & M32, attribs=(hit[1], hit[2]))
                elif op == 98:  # OpImageRead
                    c = vals[o[3]]
                    vals[o[1]] = [F32(t) for t in drv.image_load(_signed(c[0]), _signed(c[1]))]
                elif op == 99:  # OpImageWrite
                    c = vals[o[1]]
                    drv.image_store(_signed(c[0]), _signed(c[1]), vals[o[2]])
                elif op in (246, 247):  # merge hints
                    pass
                elif op == 249:
                    label = o[0]
                    break
                elif op == 250:
                    label = o[1] if vals[o[0]] else o[2]
                    break
                elif op == 253:
                    self.n_instructions += count
                    return None
                elif op == 254:
                    self.n_instructions += count
                    return _copy(vals[o[0]])
                else:
                    raise NotImplementedError("SPIR-V opcode %d" % op)
            else:
                raise RuntimeError("block without terminator")
