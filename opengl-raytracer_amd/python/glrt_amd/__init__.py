"""MI355X-native path-tracing hot path behind tatsy/opengl-raytracer's Scene/Window surface.

  glrt_amd.device  -- ctypes binding of the C-ABI HIP layer (libglrtx.so, include/glrtx.h)
  glrt_amd.host    -- ctypes binding of the CPU host helpers (libglrt_host.so, include/glrt_host.h)
  glrt_amd.scenes  -- synthetic scenes in the reference's flat buffer wire format
"""
