#version 130
// Conformance fixture written for this repository (not a RetroArch shader): phosphor persistence
// through the PassFeedback samplers, to pin the engine's feedback binding and ping-pong swap
// (reference ShaderEngine.cpp:1285-1347, :1710-1718) against a real GL.  Intended as pass 1 of a
// two-pass preset: PassFeedback0 is the previous frame's output of pass 0, PassFeedback1 its own.
#pragma parameter PERSIST "Persistence" 0.8 0.0 1.0 0.05
#if defined(VERTEX)
in vec4 VertexCoord;
in vec4 TexCoord;
out vec2 tc;
uniform mat4 MVPMatrix;
void main()
{
    gl_Position = MVPMatrix * VertexCoord;
    tc = TexCoord.xy;
}
#elif defined(FRAGMENT)
in vec2 tc;
out vec4 FragColor;
uniform sampler2D Texture;
uniform sampler2D PassFeedback0;
uniform sampler2D PassFeedback1;
#ifdef PARAMETER_UNIFORM
uniform float PERSIST;
#else
#define PERSIST 0.8
#endif
void main()
{
    vec3 cur = texture(Texture, tc).rgb;
    vec3 old0 = texture(PassFeedback0, tc).rgb;
    vec3 old1 = texture(PassFeedback1, tc).rgb;
    FragColor = vec4(max(cur * 0.75 + old0 * 0.25, old1 * PERSIST), 1.0);
}
#endif
