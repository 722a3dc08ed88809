"""Minimal glTF 2.0 ingestion into the host mirror (rayca_amd.model.Model).

Follows what the reference's loader accepts (rayca-model/src/loader/gltf.rs:56-578): .gltf JSON with
data-URI or external .bin buffers, float VEC2/VEC3/VEC4 attributes with byteStride, u8/u16/u32
indices kept byte-packed, PBR metallic-roughness materials, perspective cameras, nodes with TRS or
a matrix.  No GLB, no buffer-view images, no lights (the reference has none of these either:
gltf.rs:313,84-99).  Textures need PNG/JPEG decoding, which this image has no library for; a model
that references image files raises NotImplementedError.

Matrix nodes: the reference calls the `gltf` crate's `Transform::decomposed()` (gltf 1.4.1,
Cargo.lock:899; call site loader/gltf.rs:530), which is not vendored under /root/reference.  Its
published algorithm is restated in `decompose_matrix` below (column lengths as scale, sign of the
determinant on z, then the trace-based matrix->quaternion conversion), evaluated in f32.  Parity for
matrix-authored nodes is therefore "unpinned" (SURVEY.md section 8c).
"""
from __future__ import annotations

import base64
import json
import os

import numpy as np

from . import abi
from .model import Camera, Image, Mesh, Model, Node, PbrMaterial, Primitive, Texture, Trs, TriangleMesh

_COMPONENT_DTYPE = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32,
                    5126: np.float32}
_TYPE_WIDTH = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT2": 4, "MAT3": 9, "MAT4": 16}


def decompose_matrix(m16):
    """gltf::scene::Transform::decomposed() for a column-major 4x4 matrix, in f32."""
    f = np.float32
    m = np.array(m16, dtype=np.float32).reshape(4, 4)  # m[c] = column c
    translation = (float(m[3][0]), float(m[3][1]), float(m[3][2]))
    x, y, z = m[0][:3].copy(), m[1][:3].copy(), m[2][:3].copy()

    def mag(v):
        return f(np.sqrt(f(f(v[0] * v[0]) + f(v[1] * v[1])) + f(v[2] * v[2])))

    det = float(np.linalg.det(np.stack([x, y, z]).astype(np.float64)))
    sx, sy = mag(x), mag(y)
    sz = f(np.sign(det) if det != 0 else 1.0) * mag(z)
    x, y, z = x * (f(1.0) / sx), y * (f(1.0) / sy), z * (f(1.0) / sz)
    # Quaternion::from_matrix (columns x, y, z); element m.c.r = column c, row r
    trace = f(f(x[0] + y[1]) + z[2])
    if trace >= 0:
        s = f(np.sqrt(f(1.0) + trace))
        w = f(0.5) * s
        s = f(0.5) / s
        q = (f(y[2] - z[1]) * s, f(z[0] - x[2]) * s, f(x[1] - y[0]) * s, w)
    elif x[0] > y[1] and x[0] > z[2]:
        s = f(np.sqrt(f(f(x[0] - y[1]) - z[2]) + f(1.0)))
        qx = f(0.5) * s
        s = f(0.5) / s
        q = (qx, f(y[0] + x[1]) * s, f(x[2] + z[0]) * s, f(y[2] - z[1]) * s)
    elif y[1] > z[2]:
        s = f(np.sqrt(f(f(y[1] - x[0]) - z[2]) + f(1.0)))
        qy = f(0.5) * s
        s = f(0.5) / s
        q = (f(y[0] + x[1]) * s, qy, f(z[1] + y[2]) * s, f(z[0] - x[2]) * s)
    else:
        s = f(np.sqrt(f(f(z[2] - x[0]) - y[1]) + f(1.0)))
        qz = f(0.5) * s
        s = f(0.5) / s
        q = (f(x[2] + z[0]) * s, f(z[1] + y[2]) * s, qz, f(x[1] - y[0]) * s)
    return Trs(translation=translation, rotation=tuple(float(v) for v in q),
               scale=(float(sx), float(sy), float(sz)))


def _load_buffers(doc, base_dir):
    out = []
    for b in doc.get("buffers", []):
        uri = b["uri"]
        prefix = "data:application/octet-stream;base64,"
        if uri.startswith(prefix):
            out.append(base64.b64decode(uri[len(prefix):]))
        else:
            with open(os.path.join(base_dir, uri), "rb") as fh:
                out.append(fh.read())
    return out


def _read_accessor(doc, buffers, index):
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMPONENT_DTYPE[acc["componentType"]])
    width = _TYPE_WIDTH[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", dt.itemsize * width)
    raw = np.frombuffer(buffers[view["buffer"]], dtype=np.uint8)
    count = acc["count"]
    rows = np.lib.stride_tricks.as_strided(raw[start:], shape=(count, dt.itemsize * width), strides=(stride, 1))
    return np.ascontiguousarray(rows).view(dt).reshape(count, width)


def load_gltf(path_or_doc, base_dir=None, image_decoder=None) -> Model:
    """Model::load_gltf_path (loader/gltf.rs:291-299).  `image_decoder(file bytes) -> (H, W, 3 | 4) uint8 array` supplies
    what this Python mirror has no library for (the C++ loader, include/rayca_gltf.hpp, decodes PNG and JPEG itself)."""
    if isinstance(path_or_doc, (str, os.PathLike)):
        base_dir = os.path.dirname(os.path.abspath(path_or_doc))
        with open(path_or_doc, "r") as fh:
            doc = json.load(fh)
    else:
        doc = path_or_doc
    if doc.get("images") and image_decoder is None:
        raise NotImplementedError("glTF textures need an image decoder; pass image_decoder= (the C++ loader decodes PNG / JPEG itself)")
    buffers = _load_buffers(doc, base_dir or ".")
    model = Model()
    # load_images / load_textures (gltf.rs:305-362): only the PNG data URI is recognised, anything else is a path
    for gi in doc.get("images", []):
        if "uri" not in gi:
            raise NotImplementedError("buffer-view images are todo!() in the reference (gltf.rs:313)")
        uri = gi["uri"]
        prefix = "data:image/png;base64,"
        if uri.startswith(prefix):
            raw = base64.b64decode(uri[len(prefix):])
        else:
            with open(os.path.join(base_dir or ".", uri), "rb") as fh:
                raw = fh.read()
        px = np.ascontiguousarray(image_decoder(raw), np.uint8)
        model.images.push(Image(px.shape[1], px.shape[0], abi.COLOR_RGBA8 if px.shape[2] == 4 else abi.COLOR_RGB8, px))
    for gt in doc.get("textures", []):
        model.textures.push(Texture(image=gt["source"]))
    # load_materials (gltf.rs:364-407); glTF defaults: color 1, metallic 1, roughness 1
    for gm in doc.get("materials", []):
        pbr = gm.get("pbrMetallicRoughness", {})

        def tex(d, key):
            return d[key]["index"] if key in d else None
        model.materials.push(PbrMaterial(color=tuple(pbr.get("baseColorFactor", [1, 1, 1, 1])),
                                         albedo=tex(pbr, "baseColorTexture"), normal=tex(gm, "normalTexture"),
                                         metallic_roughness=tex(pbr, "metallicRoughnessTexture"),
                                         metallic_factor=pbr.get("metallicFactor", 1.0),
                                         roughness_factor=pbr.get("roughnessFactor", 1.0)))
    # load_meshes / load_primitive / load_vertices (gltf.rs:409-492)
    for gmesh in doc.get("meshes", []):
        handles = []
        for gp in gmesh["primitives"]:
            if gp.get("mode", 4) != 4:
                raise ValueError("only TRIANGLES primitives (gltf.rs:417)")
            at = gp["attributes"]
            pos = _read_accessor(doc, buffers, at["POSITION"]).astype(np.float32)
            nrm = _read_accessor(doc, buffers, at["NORMAL"]).astype(np.float32) if "NORMAL" in at else None
            uv = _read_accessor(doc, buffers, at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else None
            col = None
            if "COLOR_0" in at:
                c = _read_accessor(doc, buffers, at["COLOR_0"]).astype(np.float32)
                col = np.concatenate([c, np.ones((c.shape[0], 1), np.float32)], 1) if c.shape[1] == 3 else c
            tan = bit = None
            if "TANGENT" in at:
                t4 = _read_accessor(doc, buffers, at["TANGENT"]).astype(np.float32)
                tan = t4[:, :3]
                n = nrm if nrm is not None else np.tile(np.array([[0, 0, 1]], np.float32), (pos.shape[0], 1))
                # bitangent = normal x tangent * w (gltf.rs:231-232), products rounded separately
                cr = np.stack([n[:, 1] * tan[:, 2] - n[:, 2] * tan[:, 1], n[:, 2] * tan[:, 0] - n[:, 0] * tan[:, 2],
                               n[:, 0] * tan[:, 1] - n[:, 1] * tan[:, 0]], 1).astype(np.float32)
                bit = cr * t4[:, 3:4]
            if "indices" in gp:
                idx = _read_accessor(doc, buffers, gp["indices"]).reshape(-1)
            else:
                idx = np.zeros(0, np.uint8)  # load_indices default (gltf.rs:101-103)
            g = model.geometries.push(TriangleMesh(pos, idx, colors=col, normals=nrm, tangents=tan,
                                                   bitangents=bit, uvs=uv))
            handles.append(model.primitives.push(Primitive(geometry=g, material=gp.get("material"))))
        model.meshes.push(Mesh(primitives=handles))
    # load_cameras (gltf.rs:494-518)
    for gc in doc.get("cameras", []):
        if gc["type"] == "perspective":
            model.cameras.push(Camera(yfov_radians=gc["perspective"]["yfov"]))
        else:
            model.cameras.push(Camera(yfov_radians=1.0))  # Camera::orthographic sets yfov 1.0 (camera.rs:70)
    # load_nodes / create_node (gltf.rs:520-566)
    scene_index = doc.get("scene", 0)
    model.root.children = list(doc["scenes"][scene_index]["nodes"]) if doc.get("scenes") else []
    for gn in doc.get("nodes", []):
        if "matrix" in gn:
            trs = decompose_matrix(gn["matrix"])
        else:
            trs = Trs(translation=tuple(gn.get("translation", (0, 0, 0))),
                      rotation=tuple(gn.get("rotation", (0, 0, 0, 1))), scale=tuple(gn.get("scale", (1, 1, 1))))
        model.nodes.push(Node(trs=trs, children=list(gn.get("children", [])), mesh=gn.get("mesh"),
                              camera=gn.get("camera"), name=gn.get("name", "Unknown")))
    return model
