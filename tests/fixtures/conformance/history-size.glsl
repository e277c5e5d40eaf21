#version 130
// Conformance fixture written for this repository (not a RetroArch shader): a frame-history shader that READS
// ITS SIZE UNIFORMS, to pin what the reference's history push does with them.  The push re-draws the final
// output through pass 0's program (reference ShaderEngine.cpp:1805-1834) without touching the program's
// uniforms: TextureSize / InputSize / OutputSize still hold what pass 0's own draw of the frame set - stale
// values whenever the final output's size differs from pass 0's input or output.  Intended as pass 0, at a
// geometry where those differ (a scaled pass 0, or more passes behind it).
#pragma parameter HS_MIX "History weight" 0.3 0.0 1.0 0.05
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
uniform sampler2D PrevTexture;
uniform sampler2D Prev1Texture;
uniform vec2 TextureSize;
uniform vec2 InputSize;
uniform vec2 OutputSize;
#ifdef PARAMETER_UNIFORM
uniform float HS_MIX;
#else
#define HS_MIX 0.3
#endif
void main()
{
    vec2 tx = vec2(1.0 / TextureSize.x, 0.0);                 // one texel of what pass 0 believes its input to be
    vec3 cur = texture(Texture, tc).rgb;
    vec3 right = texture(Texture, tc + tx).rgb;
    vec3 old0 = texture(PrevTexture, tc).rgb;
    vec3 old1 = texture(Prev1Texture, tc + tx).rgb;
    float row = floor(tc.y * OutputSize.y);                    // a two-row pattern in pass 0's OUTPUT rows
    float dim = (row * 0.5 - floor(row * 0.5)) < 0.25 ? 1.0 : 0.75;
    float cover = InputSize.y / TextureSize.y;                // 1 in this engine (InputSize == TextureSize)
    vec3 now = cur * 0.75 + right * 0.25;
    vec3 hist = old0 * 0.625 + old1 * 0.375;
    FragColor = vec4((now + (hist - now) * HS_MIX) * dim * cover, 1.0);
}
#endif
