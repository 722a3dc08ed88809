//! rayca_shim.rs -- the Rust side of `include/rayca_hip.h`: what a maintainer adds to rayca-soft (as `src/hip.rs`,
//! behind a `hip` feature) to render through librayca_hip.so instead of the CPU path.
//!
//! The reference has no FFI for this path; the interface replaced is
//!     pub trait Draw { fn draw(&mut self, scene: &Scene, image: &mut Image); }        rayca-soft/src/draw.rs:7-9
//! implemented by `SoftRenderer { pub config: Config }` (rayca-soft/src/scene.rs:11-14,88-154).  `HipRenderer` below has
//! the same field and the same trait: swap the type, keep the call.
//!
//! NOT COMPILED in the environment this was written in (no rustc / cargo there, no network).  What IS checked
//! mechanically: every `#[repr(C)]` struct below has the fields of its C counterpart in the same order with types of the
//! same size, and the `extern "C"` block declares exactly the header's entry points with the same number of parameters
//! (tests/test_shim.py parses both files).  `FlatScene::from` is a transliteration of `rayca_amd/model.py::flatten` and
//! `include/rayca.hpp::FlatScene`, which are tested against each other byte for byte.
//!
//! Two places need a one-line accessor in rayca-model because the fields are private there (marked `ACCESSOR` below):
//! `DirectionalLight::{color, intensity}` (light/directional.rs:7-10) -- `get_radiance()` is used instead -- and nothing else:
//! `Sphere`, `Image` and `TriangleIndices` already expose what is needed.
#![cfg(feature = "hip")]
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_void};

use crate::*;

pub const RAYCA_ABI_VERSION: u32 = 2;
pub const RAYCA_NONE: u32 = 0xFFFF_FFFF; // Handle::NONE, rayca-util/src/pack.rs:61-64

// ---- status codes ---------------------------------------------------------------------------------------------------
pub const RAYCA_OK: i32 = 0;
pub const RAYCA_ERR_BAD_ARG: i32 = -1;
pub const RAYCA_ERR_NO_CAMERA: i32 = -2; // assert!(!camera_draw_infos.is_empty())   scene.rs:109
pub const RAYCA_ERR_EMPTY_SCENE: i32 = -3; // Tlas::intersects assert                  tlas.rs:272
pub const RAYCA_ERR_HIP: i32 = -4;
pub const RAYCA_ERR_OOM: i32 = -5;
pub const RAYCA_ERR_UNSUPPORTED: i32 = -6; // todo!() arms of the reference reached by a ray
pub const RAYCA_ERR_NO_DEVICE: i32 = -7;
pub const RAYCA_ERR_BVH_DEPTH: i32 = -8;
pub const RAYCA_ERR_RCCL: i32 = -9;

// ---- enums crossing the ABI as u32 (the reference's own #[repr(u32)] orders: `as u32` is the conversion) -----------------
pub const RAYCA_MATERIAL_PBR: u32 = 0;
pub const RAYCA_MATERIAL_PHONG: u32 = 1;
pub const RAYCA_MATERIAL_GGX: u32 = 2;
pub const RAYCA_LIGHT_DIRECTIONAL: u32 = 0;
pub const RAYCA_LIGHT_POINT: u32 = 1;
pub const RAYCA_LIGHT_QUAD: u32 = 2;
pub const RAYCA_GEOMETRY_TRIANGLE_MESH: u32 = 0;
pub const RAYCA_GEOMETRY_SPHERE: u32 = 1;
pub const RAYCA_COLOR_RGB8: u32 = 0;
pub const RAYCA_COLOR_RGBA8: u32 = 1;
pub const RAYCA_COLOR_RGBA32F: u32 = 2;
pub const RAYCA_BUILDER_REFERENCE: u32 = 0;
pub const RAYCA_BUILDER_SAH: u32 = 1;
pub const RAYCA_TRAVERSAL_ORDERED: u32 = 0;
pub const RAYCA_TRAVERSAL_EXHAUSTIVE: u32 = 1;
pub const RAYCA_ENGINE_AUTO: u32 = 0;
pub const RAYCA_ENGINE_GENERAL: u32 = 1;
pub const RAYCA_ENGINE_WAVEFRONT: u32 = 2;
pub const RAYCA_ENGINE_FUSED: u32 = 3;
pub const RAYCA_CAMERA_AUTO: u32 = 0;
pub const RAYCA_CAMERA_GENERATION: u32 = 1;
pub const RAYCA_CAMERA_REFILL: u32 = 2;
// RaycaStats.class_ms / class_launches
pub const RAYCA_KERNEL_GENERATION: u32 = 0;
pub const RAYCA_KERNEL_FLAT_REFILL: u32 = 1;
pub const RAYCA_KERNEL_WF_TRACE: u32 = 2;
pub const RAYCA_KERNEL_QUEUE_REFILL: u32 = 3;
pub const RAYCA_KERNEL_WF_SHADE: u32 = 4;
pub const RAYCA_KERNEL_WF_SHADOW: u32 = 5;
pub const RAYCA_KERNEL_SHADOW_REFILL: u32 = 6;
pub const RAYCA_KERNEL_OTHER: u32 = 7;
pub const RAYCA_KERNEL_CLASSES: u32 = 8;
pub const RAYCA_GATHER_RCCL: u32 = 0;
pub const RAYCA_GATHER_PEER_COPY: u32 = 1;

// ---- structs, field for field as in include/rayca_hip.h ---------------------------------------------------------------------
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaConfig {
    pub bvh: u32,
    pub light_samples: u32,
    pub light_stratify: u32,
    pub samples_per_pixel: u32,
    pub russian_roulette: u32,
    pub direct_sampler: u32,
    pub indirect_sampler: u32,
    pub integrator: u32,
    pub max_depth: u32,
    pub gamma: f32,
    pub seed: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaTrs {
    pub translation: [f32; 3],
    pub rotation: [f32; 4],
    pub scale: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RaycaNode {
    pub parent: i32,
    pub model: u32,
    pub mesh: u32,
    pub camera: u32,
    pub light: u32,
    pub trs: RaycaTrs,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaMesh {
    pub first_primitive: u32,
    pub primitive_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaPrimitive {
    pub geometry: u32,
    pub material: u32,
    pub first_vertex: u32,
    pub vertex_count: u32,
    pub index_byte_offset: u64,
    pub index_count: u32,
    pub index_type: u32,
    pub sphere_center: [f32; 3],
    pub sphere_radius: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaMaterial {
    pub kind: u32,
    pub albedo_texture: u32,
    pub normal_texture: u32,
    pub metallic_roughness_texture: u32,
    pub color: [f32; 4],
    pub metallic_factor: f32,
    pub roughness_factor: f32,
    pub shininess: f32,
    pub pad0: f32,
    pub ambient: [f32; 4],
    pub emission: [f32; 4],
    pub diffuse: [f32; 4],
    pub specular: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaTexture {
    pub image: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaImage {
    pub width: u32,
    pub height: u32,
    pub color_type: u32,
    pub pad0: u32,
    pub byte_offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaCamera {
    pub yfov_radians: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaLight {
    pub kind: u32,
    pub material: u32,
    pub intensity: f32,
    pub pad0: f32,
    pub color: [f32; 4],
    pub attenuation: [f32; 3],
    pub pad1: f32,
    pub ab: [f32; 3],
    pub pad2: f32,
    pub ac: [f32; 3],
    pub pad3: f32,
}

#[repr(C)]
pub struct RaycaSceneDesc {
    pub abi_version: u32,
    pub flags: u32,
    pub nodes: *const RaycaNode,
    pub node_count: u32,
    pub meshes: *const RaycaMesh,
    pub mesh_count: u32,
    pub primitives: *const RaycaPrimitive,
    pub primitive_count: u32,
    pub vertex_count: u32,
    pub positions: *const f32,
    pub colors: *const f32,
    pub normals: *const f32,
    pub tangents: *const f32,
    pub bitangents: *const f32,
    pub uvs: *const f32,
    pub index_bytes: *const u8,
    pub index_byte_count: u64,
    pub materials: *const RaycaMaterial,
    pub material_count: u32,
    pub textures: *const RaycaTexture,
    pub texture_count: u32,
    pub images: *const RaycaImage,
    pub image_count: u32,
    pub image_bytes: *const u8,
    pub image_byte_count: u64,
    pub cameras: *const RaycaCamera,
    pub camera_count: u32,
    pub lights: *const RaycaLight,
    pub light_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaBuildOptions {
    pub builder: u32,
    pub device: u32,
    pub build_on_host: u32,
    pub reserved: [u32; 5],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaTile {
    pub part: u32,
    pub parts: u32,
    pub band_rows: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RaycaRenderOptions {
    pub traversal: u32,
    pub collect_stats: u32,
    pub tile: RaycaTile,
    pub stream: *mut c_void,
    pub engine: u32,
    pub context: u32,
    pub camera_rays: u32,
    pub reserved: u32,
    pub wait_event: *mut c_void,
    pub record_event: *mut c_void,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaMultiOptions {
    pub traversal: u32,
    pub collect_stats: u32,
    pub band_rows: u32,
    pub gather: u32,
    pub engine: u32,
    pub output_on_device: u32,
    pub context: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaStats {
    pub rays_primary: u64,
    pub rays_shadow: u64,
    pub rays_bounce: u64,
    pub boxes_tested: u64,
    pub triangles_tested: u64,
    pub hits_shaded: u64,
    pub wave_box_slots: u64,
    pub wave_triangle_slots: u64,
    pub kernel_ms: f32,
    pub trace_kernel_ms: f32,
    pub kernel_launches: u32,
    pub trace_kernel_launches: u32,
    pub rows_rendered: u32,
    pub node_format: u32,
    pub class_ms: [f32; 8],
    pub class_launches: [u32; 8],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RaycaSceneInfo {
    pub triangle_count: u32,
    pub sphere_count: u32,
    pub blas_count: u32,
    pub node_count: u32,
    pub max_depth: u32,
    pub light_count: u32,
    pub device_bytes: u64,
    pub build_ms: f32,
    pub runtime_init_ms: f32,
}

/// opaque: owns the device-resident scene + BVH
#[repr(C)]
pub struct RaycaScene {
    _private: [u8; 0],
}

extern "C" {
    pub fn rayca_hip_version() -> u32;
    pub fn rayca_hip_device_count() -> i32;
    pub fn rayca_hip_selftest() -> i32;
    pub fn rayca_hip_last_error(buf: *mut c_char, len: usize);
    pub fn rayca_hip_config_default(out: *mut RaycaConfig);
    pub fn rayca_hip_scene_create(desc: *const RaycaSceneDesc, cfg: *const RaycaConfig, opts: *const RaycaBuildOptions, out: *mut *mut RaycaScene) -> i32;
    pub fn rayca_hip_scene_destroy(scene: *mut RaycaScene) -> i32;
    pub fn rayca_hip_scene_reap() -> i32;
    pub fn rayca_hip_scene_info(scene: *const RaycaScene, out: *mut RaycaSceneInfo) -> i32;
    pub fn rayca_hip_scene_finish(scene: *mut RaycaScene) -> i32;
    pub fn rayca_hip_render(scene: *mut RaycaScene, cfg: *const RaycaConfig, width: u32, height: u32, opts: *const RaycaRenderOptions, rgba8_out: *mut u8, rgba32f_out: *mut f32, stats_out: *mut RaycaStats) -> i32;
    pub fn rayca_hip_render_device(scene: *mut RaycaScene, cfg: *const RaycaConfig, width: u32, height: u32, opts: *const RaycaRenderOptions, d_rgba8_out: *mut c_void, d_rgba32f_out: *mut c_void, stats_out: *mut RaycaStats) -> i32;
    pub fn rayca_hip_render_multi(scenes: *const *mut RaycaScene, count: u32, cfg: *const RaycaConfig, width: u32, height: u32, opts: *const RaycaMultiOptions, rgba8_out: *mut c_void, stats_out: *mut RaycaStats) -> i32;
    pub fn rayca_hip_render_multi_issue(scenes: *const *mut RaycaScene, count: u32, cfg: *const RaycaConfig, width: u32, height: u32, opts: *const RaycaMultiOptions, rgba8_out: *mut c_void) -> i32;
    pub fn rayca_hip_render_multi_wait(scenes: *const *mut RaycaScene, count: u32, context: u32) -> i32;
    pub fn rayca_hip_rccl_status() -> i32;
    pub fn rayca_hip_tile_rows(tile: *const RaycaTile, height: u32) -> u32;
    pub fn rayca_hip_trace_rays(scene: *mut RaycaScene, opts: *const RaycaRenderOptions, count: u32, rays: *const f32, t_out: *mut f32, prim_out: *mut u32, uv_out: *mut f32, stats_out: *mut RaycaStats) -> i32;
    pub fn rayca_hip_scene_primitive_order(scene: *const RaycaScene, prim_order: *mut u32, capacity: u32) -> i32;
    pub fn rayca_hip_scene_read_nodes(scene: *mut RaycaScene, which: u32, out: *mut c_void, capacity_bytes: u64, bytes_out: *mut u64) -> i32;
}

// ---- Config -------------------------------------------------------------------------------------------------------------
impl From<&Config> for RaycaConfig {
    fn from(c: &Config) -> Self {
        RaycaConfig {
            bvh: c.bvh as u32,
            light_samples: c.light_samples,
            light_stratify: c.light_stratify as u32,
            samples_per_pixel: c.samples_per_pixel,
            russian_roulette: c.russian_roulette as u32,
            direct_sampler: c.direct_sampler as u32,     // #[repr(u32)]  sampler/mod.rs:41-50
            indirect_sampler: c.indirect_sampler as u32,
            integrator: c.integrator as u32,             // #[repr(u32)]  integrator/mod.rs:32-41
            max_depth: c.max_depth,
            gamma: c.gamma,
            seed: 0, // counter-based RNG key: no reference counterpart (the reference draws from an OS-seeded fastrand)
            reserved: 0,
        }
    }
}

// ---- &Scene -> RaycaSceneDesc ---------------------------------------------------------------------------------------------
/// Owns the arrays a `RaycaSceneDesc` points into.  Serialisation only -- world transforms, BVH, normal matrices are
/// computed inside the library in the reference's own operation order.  The walk is `SceneDrawInfo::traverse_scene`
/// (rayca-soft/src/scene.rs:206-282): DFS pre-order, a scene node's model (its root, then its subtree) before the node's
/// own children.
#[derive(Default)]
pub struct FlatScene {
    pub nodes: Vec<RaycaNode>,
    pub meshes: Vec<RaycaMesh>,
    pub primitives: Vec<RaycaPrimitive>,
    pub positions: Vec<f32>,
    pub colors: Vec<f32>,
    pub normals: Vec<f32>,
    pub tangents: Vec<f32>,
    pub bitangents: Vec<f32>,
    pub uvs: Vec<f32>,
    pub index_bytes: Vec<u8>,
    pub materials: Vec<RaycaMaterial>,
    pub textures: Vec<RaycaTexture>,
    pub images: Vec<RaycaImage>,
    pub image_bytes: Vec<u8>,
    pub cameras: Vec<RaycaCamera>,
    pub lights: Vec<RaycaLight>,
}

/// where one model's packs start in the global arrays (a model is emitted once, however many scene nodes use it)
#[derive(Clone, Copy, Default)]
struct ModelBase {
    done: bool,
    mesh: u32,
    material: u32,
    texture: u32,
    image: u32,
    camera: u32,
    light: u32,
}

fn trs_to_abi(t: &Trs) -> RaycaTrs {
    RaycaTrs {
        translation: [t.translation.get_x(), t.translation.get_y(), t.translation.get_z()],
        rotation: [t.rotation.get_x(), t.rotation.get_y(), t.rotation.get_z(), t.rotation.get_w()],
        scale: [t.scale.get_x(), t.scale.get_y(), t.scale.get_z()],
    }
}
fn rgba(c: &Color) -> [f32; 4] {
    [c.r, c.g, c.b, c.a]
}
fn xyz(v: &Vec3) -> [f32; 3] {
    [v.get_x(), v.get_y(), v.get_z()]
}
fn handle_or_none<T>(h: Handle<T>, base: u32) -> u32 {
    if h.is_valid() { base + h.id } else { RAYCA_NONE }
}
fn opt_handle<T>(h: Option<Handle<T>>, base: Option<u32>) -> u32 {
    match (h, base) {
        (Some(h), Some(b)) if h.is_valid() => b + h.id,
        _ => RAYCA_NONE,
    }
}
/// one entry per handle ID (not per stored element): `base + handle.id` must stay a valid index after a `Pack::remove`
fn for_each_id<T>(pack: &Pack<T>, mut f: impl FnMut(Option<&T>)) {
    for id in 0..pack.get_indices().len() as u32 {
        f(pack.get(Handle::new(id)));
    }
}

impl FlatScene {
    fn emit(&mut self, trs: &Trs, parent: i32, model: u32, payload: Option<&Node>, base: Option<&ModelBase>) -> u32 {
        let (mut mesh, mut camera, mut light) = (RAYCA_NONE, RAYCA_NONE, RAYCA_NONE);
        if let (Some(n), Some(b)) = (payload, base) {
            mesh = opt_handle(n.mesh, Some(b.mesh));
            camera = opt_handle(n.camera, Some(b.camera));
            light = opt_handle(n.light, Some(b.light));
        }
        self.nodes.push(RaycaNode { parent, model, mesh, camera, light, trs: trs_to_abi(trs) });
        self.nodes.len() as u32 - 1
    }

    fn material_to_abi(model: &Model, m: &Material, tex_base: u32) -> RaycaMaterial {
        let black = [0.0, 0.0, 0.0, 1.0];
        let mut r = RaycaMaterial {
            albedo_texture: RAYCA_NONE,
            normal_texture: RAYCA_NONE,
            metallic_roughness_texture: RAYCA_NONE,
            color: [1.0; 4],
            ambient: black,
            emission: black,
            diffuse: black,
            specular: black,
            roughness_factor: 1.0,
            ..Default::default()
        };
        match m {
            // Material::Pbr(Handle::NONE) is Material::DEFAULT -> PbrMaterial::WHITE (material/mod.rs:22-29)
            Material::Pbr(h) => {
                r.kind = RAYCA_MATERIAL_PBR;
                if let Some(p) = model.pbr_materials.get(*h) {
                    r.color = rgba(&p.color);
                    r.albedo_texture = handle_or_none(p.albedo, tex_base);
                    r.normal_texture = handle_or_none(p.normal, tex_base);
                    r.metallic_roughness_texture = handle_or_none(p.metallic_roughness, tex_base);
                    r.metallic_factor = p.metallic_factor;
                    r.roughness_factor = p.roughness_factor;
                }
            }
            Material::Phong(h) => {
                r.kind = RAYCA_MATERIAL_PHONG;
                if let Some(p) = model.phong_materials.get(*h) {
                    r.ambient = rgba(&p.ambient);
                    r.emission = rgba(&p.emission);
                    r.diffuse = rgba(&p.diffuse);
                    r.specular = rgba(&p.specular);
                    r.shininess = p.shininess;
                }
            }
            Material::Ggx(h) => {
                r.kind = RAYCA_MATERIAL_GGX;
                if let Some(g) = model.ggx_materials.get(*h) {
                    r.diffuse = rgba(&g.diffuse);
                    r.specular = rgba(&g.specular);
                    r.roughness_factor = g.roughness;
                }
            }
        }
        r
    }

    fn add_model_payload(&mut self, model: &Model, base: &mut ModelBase) {
        if base.done {
            return;
        }
        base.mesh = self.meshes.len() as u32;
        base.material = self.materials.len() as u32;
        base.texture = self.textures.len() as u32;
        base.image = self.images.len() as u32;
        base.camera = self.cameras.len() as u32;
        base.light = self.lights.len() as u32;
        let b = *base;

        for_each_id(&model.images, |im| {
            let mut r = RaycaImage { byte_offset: self.image_bytes.len() as u64, ..Default::default() };
            if let Some(im) = im {
                r.width = im.width();
                r.height = im.height();
                r.color_type = im.color_type as u32; // ColorType: RGB8, RGBA8, RGBA32F  rayca-math/src/color/mod.rs:19-25
                self.image_bytes.extend_from_slice(im.bytes());
            }
            self.images.push(r);
        });
        for_each_id(&model.textures, |t| {
            self.textures.push(RaycaTexture { image: t.map_or(RAYCA_NONE, |t| handle_or_none(t.image, b.image)) });
        });
        for_each_id(&model.materials, |m| {
            let m = m.copied().unwrap_or(Material::DEFAULT);
            self.materials.push(Self::material_to_abi(model, &m, b.texture));
        });
        for_each_id(&model.cameras, |c| {
            self.cameras.push(RaycaCamera { yfov_radians: c.map_or(0.0, |c| c.yfov_radians) });
        });
        for_each_id(&model.lights, |l| {
            let mut r = RaycaLight { material: RAYCA_NONE, attenuation: [0.0, 0.0, 1.0], ..Default::default() };
            match l {
                Some(Light::Point(p)) => {
                    r.kind = RAYCA_LIGHT_POINT;
                    r.intensity = p.intensity;
                    r.color = rgba(&p.color);
                    r.attenuation = xyz(&p.attenuation);
                }
                Some(Light::Quad(q)) => {
                    r.kind = RAYCA_LIGHT_QUAD;
                    r.intensity = q.intensity;
                    r.color = rgba(&q.color);
                    r.ab = xyz(&q.ab);
                    r.ac = xyz(&q.ac);
                    r.material = handle_or_none(q.material, b.material);
                }
                Some(Light::Directional(d)) => {
                    // ACCESSOR: DirectionalLight's fields are private; get_radiance() = intensity * color (alpha 1)
                    let rad = d.get_radiance();
                    r.kind = RAYCA_LIGHT_DIRECTIONAL;
                    r.intensity = 1.0;
                    r.color = [rad.get_x(), rad.get_y(), rad.get_z(), 1.0];
                }
                None => {}
            }
            self.lights.push(r);
        });
        // a Mesh is a list of primitive handles (mesh.rs:32-35): its primitives are emitted as one contiguous range
        for_each_id(&model.meshes, |mesh| {
            let first = self.primitives.len() as u32;
            let handles: &[Handle<Primitive>] = mesh.map_or(&[], |m| &m.primitives);
            for ph in handles {
                let p = model.primitives.get(*ph).expect("mesh refers to a primitive that does not exist");
                let mut rp = RaycaPrimitive { material: handle_or_none(p.material, b.material), ..Default::default() };
                match model.geometries.get(p.geometry).expect("primitive refers to a geometry that does not exist") {
                    Geometry::Sphere(s) => {
                        rp.geometry = RAYCA_GEOMETRY_SPHERE;
                        let c = s.get_model_center();
                        rp.sphere_center = [c.get_x(), c.get_y(), c.get_z()];
                        rp.sphere_radius = s.get_model_radius();
                        rp.index_type = ComponentType::U32 as u32;
                    }
                    Geometry::TriangleMesh(t) => {
                        rp.geometry = RAYCA_GEOMETRY_TRIANGLE_MESH;
                        rp.first_vertex = (self.positions.len() / 3) as u32;
                        rp.vertex_count = t.vertices.len() as u32;
                        rp.index_type = t.indices.index_type as u32; // glTF numbers 5121 / 5123 / 5125  triangle.rs:180-201
                        while self.index_bytes.len() % 4 != 0 {
                            self.index_bytes.push(0);
                        }
                        rp.index_byte_offset = self.index_bytes.len() as u64;
                        rp.index_count = t.indices.get_index_count() as u32;
                        // the byte-packed indices verbatim (triangle.rs:215-307)
                        match t.indices.index_type {
                            ComponentType::U8 => self.index_bytes.extend_from_slice(t.indices.get_indices::<u8>()),
                            ComponentType::U16 => t.indices.get_indices::<u16>().iter().for_each(|i| self.index_bytes.extend_from_slice(&i.to_ne_bytes())),
                            ComponentType::U32 => t.indices.get_indices::<u32>().iter().for_each(|i| self.index_bytes.extend_from_slice(&i.to_ne_bytes())),
                            other => panic!("Unsupported index type: {:?}", other), // primitive.rs:258
                        }
                        for v in &t.vertices {
                            self.positions.extend_from_slice(&[v.pos.get_x(), v.pos.get_y(), v.pos.get_z()]);
                            self.colors.extend_from_slice(&rgba(&v.ext.color));
                            self.normals.extend_from_slice(&xyz(&v.ext.normal));
                            self.tangents.extend_from_slice(&xyz(&v.ext.tangent));
                            self.bitangents.extend_from_slice(&xyz(&v.ext.bitangent));
                            self.uvs.extend_from_slice(&[v.ext.uv.x, v.ext.uv.y]);
                        }
                    }
                }
                self.primitives.push(rp);
            }
            self.meshes.push(RaycaMesh { first_primitive: first, primitive_count: handles.len() as u32 });
        });
        base.done = true;
    }

    fn walk_model_node(&mut self, model: &Model, model_id: u32, nh: Handle<Node>, parent: i32, base: &ModelBase) {
        let node = model.nodes.get(nh).expect("model node handle out of range");
        let me = self.emit(&node.trs, parent, model_id, Some(node), Some(base));
        for c in &node.children {
            self.walk_model_node(model, model_id, *c, me as i32, base);
        }
    }

    fn walk_scene_node(&mut self, scene: &Scene, nh: Handle<Node>, parent: i32, bases: &mut Vec<ModelBase>) {
        let node = scene.nodes.get(nh).expect("scene node handle out of range");
        let me = self.emit(&node.trs, parent, RAYCA_NONE, None, None);
        if let Some(mh) = node.model {
            let model = scene.models.get(mh).expect("scene node refers to a model that does not exist");
            let mut base = bases[mh.id as usize];
            self.add_model_payload(model, &mut base);
            bases[mh.id as usize] = base;
            let mroot = self.emit(&model.root.trs, me as i32, mh.id, None, None);
            for c in &model.root.children {
                self.walk_model_node(model, mh.id, *c, mroot as i32, &base);
            }
        }
        for c in &node.children {
            self.walk_scene_node(scene, *c, me as i32, bases);
        }
    }

    /// The descriptor borrows from `self`: keep the FlatScene alive until rayca_hip_scene_create has returned.
    pub fn desc(&self) -> RaycaSceneDesc {
        RaycaSceneDesc {
            abi_version: RAYCA_ABI_VERSION,
            flags: 0,
            nodes: self.nodes.as_ptr(),
            node_count: self.nodes.len() as u32,
            meshes: self.meshes.as_ptr(),
            mesh_count: self.meshes.len() as u32,
            primitives: self.primitives.as_ptr(),
            primitive_count: self.primitives.len() as u32,
            vertex_count: (self.positions.len() / 3) as u32,
            positions: self.positions.as_ptr(),
            colors: self.colors.as_ptr(),
            normals: self.normals.as_ptr(),
            tangents: self.tangents.as_ptr(),
            bitangents: self.bitangents.as_ptr(),
            uvs: self.uvs.as_ptr(),
            index_bytes: self.index_bytes.as_ptr(),
            index_byte_count: self.index_bytes.len() as u64,
            materials: self.materials.as_ptr(),
            material_count: self.materials.len() as u32,
            textures: self.textures.as_ptr(),
            texture_count: self.textures.len() as u32,
            images: self.images.as_ptr(),
            image_count: self.images.len() as u32,
            image_bytes: self.image_bytes.as_ptr(),
            image_byte_count: self.image_bytes.len() as u64,
            cameras: self.cameras.as_ptr(),
            camera_count: self.cameras.len() as u32,
            lights: self.lights.as_ptr(),
            light_count: self.lights.len() as u32,
        }
    }
}

impl From<&Scene> for FlatScene {
    fn from(scene: &Scene) -> Self {
        let mut flat = FlatScene::default();
        let mut bases = vec![ModelBase::default(); scene.models.get_indices().len()];
        let root = flat.emit(&scene.root.trs, -1, RAYCA_NONE, None, None);
        for c in &scene.root.children {
            flat.walk_scene_node(scene, *c, root as i32, &mut bases);
        }
        flat
    }
}

// ---- the renderer ---------------------------------------------------------------------------------------------------------
fn check(rc: i32) {
    if rc != RAYCA_OK {
        let mut buf = [0 as c_char; 512];
        unsafe { rayca_hip_last_error(buf.as_mut_ptr(), buf.len()) };
        // the reference panics in the same situations: no camera scene.rs:109, empty TLAS tlas.rs:272, todo!() arms
        panic!("rayca_hip error {rc}: {}", unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy());
    }
}

/// A scene kept resident on the device (no reference counterpart: `SoftRenderer::draw` rebuilds SceneDrawInfo, BvhScene
/// and the Tlas on every call, scene.rs:90-99; a viewer keeps one of these and calls `draw` per frame).
pub struct DeviceScene {
    handle: *mut RaycaScene,
}

impl DeviceScene {
    pub fn new(scene: &Scene, config: &Config, builder: u32, device: u32) -> Self {
        let flat = FlatScene::from(scene);
        let cfg = RaycaConfig::from(config);
        let opts = RaycaBuildOptions { builder, device, ..Default::default() };
        let mut handle: *mut RaycaScene = std::ptr::null_mut();
        check(unsafe { rayca_hip_scene_create(&flat.desc(), &cfg, &opts, &mut handle) });
        Self { handle }
    }

    pub fn draw(&mut self, config: &Config, image: &mut Image) -> RaycaStats {
        let cfg = RaycaConfig::from(config);
        let mut stats = RaycaStats::default();
        let (w, h) = (image.width(), image.height());
        check(unsafe { rayca_hip_render(self.handle, &cfg, w, h, std::ptr::null(), image.bytes_mut().as_mut_ptr(), std::ptr::null_mut(), &mut stats) });
        stats
    }
}

impl Drop for DeviceScene {
    fn drop(&mut self) {
        unsafe { rayca_hip_scene_destroy(self.handle) };
    }
}

/// Drop-in for `SoftRenderer`: same field, same trait.
#[derive(Default)]
pub struct HipRenderer {
    pub config: Config,
}

impl HipRenderer {
    pub fn new_with_config(config: Config) -> Self {
        Self { config }
    }
}

impl Draw for HipRenderer {
    /// Like the reference, every call flattens the scene, builds the acceleration structure and drops it; the tree is the
    /// reference's own (RAYCA_BUILDER_REFERENCE), so depth ties resolve as they do there.
    fn draw(&mut self, scene: &Scene, image: &mut Image) {
        assert!(image.color_type == ColorType::RGBA8, "draw() writes RGBA8 images (scene.rs:117)");
        let mut resident = DeviceScene::new(scene, &self.config, RAYCA_BUILDER_REFERENCE, 0);
        resident.draw(&self.config, image);
    }
}
