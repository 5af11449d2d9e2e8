"""Scene I/O ("next" row N4 of SURVEY §8f): 3DGS-compatible binary PLY import/export and the checkpoint dictionary
keys of the reference.

PLY layout (threedgrut/export/ply_exporter.py:34-84, model/model.py:671-719): per vertex, little-endian float32
    x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3
with PRE-activation values (density logit, log-scale, un-normalised wxyz quaternion) and the specular SH stored
CHANNEL-major (f_rest = [R coeffs 1..15 | G ... | B ...]) while the tracer's [N,48] tensor is COEFFICIENT-major
(coefficient k, channel c at 3k+c) — the transpose is the only non-trivial part.  Written with numpy only (the
reference depends on the `plyfile` package, which is not available here).
"""
import numpy as np

_N_SPEC = 15


def _attribute_names():
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * _N_SPEC)]
    return names + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]


def write_ply(path, positions, density_logit, rotation_raw, log_scale, features48):
    """All arrays pre-activation, features48 coefficient-major [N,48]."""
    n = positions.shape[0]
    f = np.asarray(features48, np.float32).reshape(n, 16, 3)
    albedo = f[:, 0, :]
    spec_channel_major = f[:, 1:, :].transpose(0, 2, 1).reshape(n, 3 * _N_SPEC)
    normals = np.tile(np.array([[0, 0, 1]], np.float32), (n, 1))
    table = np.concatenate([np.asarray(positions, np.float32), normals, albedo, spec_channel_major,
                            np.asarray(density_logit, np.float32).reshape(n, 1), np.asarray(log_scale, np.float32),
                            np.asarray(rotation_raw, np.float32)], axis=1).astype("<f4")
    names = _attribute_names()
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {a}\n" for a in names) + "end_header\n"
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(np.ascontiguousarray(table).tobytes())


def read_ply(path):
    """Returns dict(positions, density_logit[N,1], rotation_raw[N,4], log_scale[N,3], features48[N,48]) (pre-activation)."""
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, n, props, in_vertex = None, None, [], False
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties are not supported on the vertex element")
                props.append((tok[2], {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1",
                                       "uint8": "u1", "int": "<i4", "int32": "<i4", "short": "<i2", "ushort": "<u2"}[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError(f"unsupported PLY format {fmt!r} (binary_little_endian only)")
        data = np.frombuffer(fh.read(n * np.dtype(props).itemsize), dtype=np.dtype(props), count=n)
    col = lambda k: np.asarray(data[k], np.float32)
    rest = sorted((p for p, _ in props if p.startswith("f_rest_")), key=lambda s: int(s.split("_")[-1]))
    if len(rest) != 3 * _N_SPEC:
        raise ValueError(f"expected {3 * _N_SPEC} f_rest_* properties (SH degree 3), found {len(rest)}")
    spec = np.stack([col(k) for k in rest], 1).reshape(n, 3, _N_SPEC).transpose(0, 2, 1)  # -> coefficient-major
    feats = np.concatenate([np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], 1)[:, None, :], spec], 1).reshape(n, 48)
    scl = sorted((p for p, _ in props if p.startswith("scale_")), key=lambda s: int(s.split("_")[-1]))
    rot = sorted((p for p, _ in props if p.startswith("rot")), key=lambda s: int(s.split("_")[-1]))
    return dict(positions=np.stack([col("x"), col("y"), col("z")], 1), density_logit=col("opacity")[:, None],
                rotation_raw=np.stack([col(k) for k in rot], 1), log_scale=np.stack([col(k) for k in scl], 1),
                features48=np.ascontiguousarray(feats, np.float32))


def scene_from_ply(path):
    """Activated scene dict (the form scenes.py generators return)."""
    d = read_ply(path)
    q = d["rotation_raw"] / np.maximum(np.linalg.norm(d["rotation_raw"], axis=1, keepdims=True), 1e-12)
    return dict(positions=d["positions"], rotation=q.astype(np.float32), scale=np.exp(d["log_scale"]).astype(np.float32),
                density=(1.0 / (1.0 + np.exp(-d["density_logit"]))).astype(np.float32), features=d["features48"])


def export_native_model(model, path):
    raw = model.raw.detach().cpu().numpy()
    write_ply(path, raw[:, 0:3], raw[:, 3:4], raw[:, 4:8], raw[:, 8:11], model.features.detach().cpu().numpy())


def checkpoint_dict(model, extra=None):
    """Parameter entries of the reference's checkpoint (model/model.py:107-134 get_model_parameters)."""
    raw = model.raw.detach()
    d = {"positions": raw[:, 0:3].clone(), "rotation": raw[:, 4:8].clone(), "scale": raw[:, 8:11].clone(),
         "density": raw[:, 3:4].clone(), "features_albedo": model.features[:, :3].detach().clone(),
         "features_specular": model.features[:, 3:].detach().clone(), "n_active_features": model.n_active_features,
         "max_n_features": model.max_n_features}
    d.update(extra or {})
    return d
