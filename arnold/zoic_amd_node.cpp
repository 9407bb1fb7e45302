// arnold/zoic_amd_node.cpp -- the Arnold-loadable camera node "zoic" backed by libzoic_amd.so.
//
// Built only where an Arnold 5 SDK exists (arnold/Makefile: `make ARNOLD_SDK=/path/to/Arnold-5.x`); this image has none,
// so the file is compile-guarded and has never been linked against a real SDK -- the macro spellings below are Arnold 5's
// public ones (ai_nodes.h / ai_cameras.h) and must be re-checked against the SDK it is first built with.
//
// It maps the reference's method table (AI_CAMERA_NODE_EXPORT_METHODS(zoicMethods), zoic.cpp:61) one to one onto the C-ABI
// of include/zoic_amd.h -- no optics here:
//   node_parameters     zoic.cpp:1547-1562  -> the same 14 AiParameter* declarations (names, types, defaults)
//   node_initialize     zoic.cpp:1565-1572  -> zoic_camera_create
//   node_update         zoic.cpp:1575-1720  -> AiTextureLoad + zoic_camera_set_bokeh_image + zoic_camera_update
//   node_finish         zoic.cpp:1723-1749  -> zoic_camera_get_counters + zoic_camera_destroy
//   camera_create_ray   zoic.cpp:1752-1990  -> zoic_camera_create_ray(cam, input, output, tid)   (re-entrant per tid)
//   camera_reverse_ray  zoic.cpp:1992-1995  -> zoic_camera_reverse_ray                           (false)
//   NodeLoader          zoic.cpp:1999-2007  -> same fields
// The per-sample callback goes through the library's resident mailbox kernel (no launch per call: ~7.5 us per sample with one
// render thread, ~0.8 M calls/s with 16 -- still ~20x below 16 threads of the CPU plug-in); a renderer that can hand over a
// bucket's samples at once should call zoic_create_rays_arnold / zoic_create_rays_device instead.
#ifdef ARNOLD_SDK

#include <ai.h>

#include <cstring>
#include <vector>

#include "zoic_amd.h"

AI_CAMERA_NODE_EXPORT_METHODS(zoicAmdMethods)

namespace {

struct NodeData { zoic_camera *cam = nullptr; };

const char *kLensModelNames[] = {"THINLENS", "RAYTRACED", NULL};   // enum LensModel, zoic.cpp:84-88

int device_from_env()
{
    const char *e = getenv("ZOIC_AMD_DEVICE");
    return e ? atoi(e) : 0;
}

}  // namespace

node_parameters
{
    AiParameterFlt("sensorWidth", 3.6f);                 // zoic.cpp:1549
    AiParameterFlt("sensorHeight", 2.4f);
    AiParameterFlt("focalLength", 2.0f);
    AiParameterFlt("fStop", 4.0f);
    AiParameterFlt("focalDistance", 100.0f);
    AiParameterBool("useImage", false);
    AiParameterStr("bokehPath", "");
    AiParameterEnum("lensModel", ZOIC_RAYTRACED, kLensModelNames);
    AiParameterStr("lensDataPath", "");
    AiParameterBool("kolbSamplingLUT", true);
    AiParameterBool("useDof", true);
    AiParameterFlt("opticalVignettingDistance", 0.0f);
    AiParameterFlt("opticalVignettingRadius", 1.0f);
    AiParameterFlt("exposureControl", 0.0f);             // zoic.cpp:1562
}

node_initialize
{
    AiCameraInitialize(node);
    NodeData *data = new NodeData();
    if (zoic_abi_version() != ZOIC_AMD_ABI_VERSION) {   // a shim built against another header must not touch the library's structs
        AiMsgError("[ZOIC] libzoic_amd.so has ABI %d, this node was built for %d", zoic_abi_version(), ZOIC_AMD_ABI_VERSION);
        AiRenderAbort();
    } else if (zoic_camera_create(device_from_env(), &data->cam) != ZOIC_OK) {
        AiMsgError("[ZOIC] %s", zoic_last_error_string());   // no gfx950 device: the library has no CPU path
        AiRenderAbort();
    }
    AiNodeSetLocalData(node, data);
}

node_update
{
    AiCameraUpdate(node, false);
    NodeData *data = static_cast<NodeData *>(AiNodeGetLocalData(node));
    if (!data || !data->cam) return;
    zoic_params p;
    zoic_params_default(&p);
    p.sensorWidth = AiNodeGetFlt(node, "sensorWidth");
    p.sensorHeight = AiNodeGetFlt(node, "sensorHeight");
    p.focalLength = AiNodeGetFlt(node, "focalLength");
    p.fStop = AiNodeGetFlt(node, "fStop");
    p.focalDistance = AiNodeGetFlt(node, "focalDistance");
    p.useImage = AiNodeGetBool(node, "useImage") ? 1 : 0;
    p.bokehPath = AiNodeGetStr(node, "bokehPath");
    p.lensModel = AiNodeGetInt(node, "lensModel");
    p.lensDataPath = AiNodeGetStr(node, "lensDataPath");
    p.kolbSamplingLUT = AiNodeGetBool(node, "kolbSamplingLUT") ? 1 : 0;
    p.useDof = AiNodeGetBool(node, "useDof") ? 1 : 0;
    p.opticalVignettingDistance = AiNodeGetFlt(node, "opticalVignettingDistance");
    p.opticalVignettingRadius = AiNodeGetFlt(node, "opticalVignettingRadius");
    p.exposureControl = AiNodeGetFlt(node, "exposureControl");
    if (p.useImage && p.bokehPath && p.bokehPath[0]) {   // imageData::read, zoic.cpp:168-219
        unsigned int w = 0, h = 0, c = 0;
        const AtString path(p.bokehPath);
        if (AiTextureGetResolution(path, &w, &h) && AiTextureGetNumChannels(path, &c) && w && h && c) {
            std::vector<float> px(static_cast<size_t>(w) * h * c);
            if (AiTextureLoad(path, true, 0, px.data()))
                zoic_camera_set_bokeh_image(data->cam, static_cast<int>(w), static_cast<int>(h), static_cast<int>(c), px.data());
        }
    }
    if (zoic_camera_update(data->cam, &p) != ZOIC_OK) {   // zoic.cpp:1589-1592, 1639-1642: message, abort, carry on
        AiMsgError("%s", zoic_last_error_string());
        AiRenderAbort();
    }
}

node_finish
{
    NodeData *data = static_cast<NodeData *>(AiNodeGetLocalData(node));
    if (!data) return;
    if (data->cam) {
        zoic_counters c;
        if (zoic_camera_get_counters(data->cam, &c) == ZOIC_OK) {   // zoic.cpp:1729-1732
            AiMsgInfo("%-40s %12llu", "[ZOIC] Succesful rays", static_cast<unsigned long long>(c.succesRays));
            AiMsgInfo("%-40s %12llu", "[ZOIC] Vignetted rays", static_cast<unsigned long long>(c.vignettedRays));
            AiMsgInfo("%-40s %12llu", "[ZOIC] Total internal reflection cases", static_cast<unsigned long long>(c.totalInternalReflection));
        }
        zoic_camera_destroy(data->cam);
    }
    delete data;
}

camera_create_ray
{
    // AtCameraInput / AtCameraOutput and zoic_camera_input / zoic_camera_output have the same field order and sizes
    // (28 / 84 bytes: include/zoic_amd.h, static_assert in capi.cpp)
    static_assert(sizeof(AtCameraInput) == sizeof(zoic_camera_input), "AtCameraInput layout");
    static_assert(sizeof(AtCameraOutput) == sizeof(zoic_camera_output), "AtCameraOutput layout");
    const NodeData *data = static_cast<const NodeData *>(AiNodeGetLocalData(node));
    zoic_camera_create_ray(data->cam, reinterpret_cast<const zoic_camera_input *>(&input), reinterpret_cast<zoic_camera_output *>(&output), tid);
}

camera_reverse_ray
{
    const NodeData *data = static_cast<const NodeData *>(AiNodeGetLocalData(node));
    float ps[2] = {Ps.x, Ps.y};
    const zoic_vec3 po = {Po.x, Po.y, Po.z};
    return zoic_camera_reverse_ray(data ? data->cam : nullptr, &po, fov, ps, &relative_time) != 0;   // false, zoic.cpp:1992-1995
}

node_loader
{
    if (i != 0) return false;
    node->methods = zoicAmdMethods;
    node->output_type = AI_TYPE_NONE;
    node->name = "zoic";
    node->node_type = AI_NODE_CAMERA;
    strcpy(node->version, AI_VERSION);
    return true;
}

#else   // no Arnold SDK in this build environment: keep the translation unit valid and say why it is empty
extern "C" const char *zoic_amd_arnold_shim_status(void) { return "built without ARNOLD_SDK: the Arnold node is not compiled in"; }
#endif
