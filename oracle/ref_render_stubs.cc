// oracle/ref_render_stubs.cc -- TEST INFRASTRUCTURE: link-time stand-ins for the few Render:: entry points the reference's
// visualization modules (spectrogram / waterfall / lineplot: module_impl.cc) reference from their PRESENT halves
// (createPresent / present / surface set-up).  The compiled-reference checker (oracle/ref_jetstream_build.sh ->
// oracle/_ref/libref_jetstream.so) compiles those translation units in place because they also hold validate / define /
// create, and drives only the COMPUTE half (computeSubmit of module_impl_native_cpu.cc) -- the reference's render
// library (src/render: fonts, text, windows, device back ends) is not built, its dependencies are network wraps.
// Every stand-in fails loudly if it is ever reached: nothing here computes anything.
#include <cstdio>
#include <cstdlib>

#include <jetstream/render/base/buffer.hh>
#include <jetstream/render/base/kernel.hh>
#include <jetstream/render/base/surface.hh>
#include <jetstream/render/base/window.hh>
#include <jetstream/render/components/axis.hh>
#include <jetstream/render/components/text.hh>

namespace {
[[noreturn]] void unreachable(const char* what) {
    std::fprintf(stderr, "oracle/ref_render_stubs.cc: %s reached -- the compiled-reference checker has no render library\n", what);
    std::abort();
}
}  // namespace

namespace Jetstream::Render {

Result Buffer::update() { unreachable("Render::Buffer::update()"); }
Result Buffer::update(const U64&, const U64&) { unreachable("Render::Buffer::update(offset, size)"); }

void Kernel::update() { unreachable("Render::Kernel::update"); }
void Surface::clearColor(const ColorRGBA<F32>&) { unreachable("Render::Surface::clearColor"); }
Result Window::bind(const std::shared_ptr<Components::Generic>&) { unreachable("Render::Window::bind(component)"); }
Result Window::unbind(const std::shared_ptr<Components::Generic>&) { unreachable("Render::Window::unbind(component)"); }
Result Window::unbind(const std::shared_ptr<WindowAttachment>&) { unreachable("Render::Window::unbind(attachment)"); }
bool Window::hasFont(const std::string&) const { unreachable("Render::Window::hasFont"); }
const std::shared_ptr<Components::Font>& Window::font(const std::string&) const { unreachable("Render::Window::font"); }

namespace Components {

const Text::ElementConfig& Text::get(const std::string&) const { unreachable("Text::get"); }
Result Text::update(const std::string&, const ElementConfig&) { unreachable("Text::update"); }
Result Text::updatePixelSize(const Extent2D<F32>&) { unreachable("Text::updatePixelSize"); }

struct Axis::Impl {};
Axis::Axis(const Config& config) : config(config) { unreachable("Render::Components::Axis::Axis"); }
Axis::~Axis() = default;
Result Axis::create(Window*) { unreachable("Axis::create"); }
Result Axis::destroy(Window*) { unreachable("Axis::destroy"); }
Result Axis::surfaceUnderlay(Render::Surface::Config&) { unreachable("Axis::surfaceUnderlay"); }
Result Axis::surfaceOverlay(Render::Surface::Config&) { unreachable("Axis::surfaceOverlay"); }
Result Axis::present() { unreachable("Axis::present"); }
Result Axis::updatePixelSize(const Extent2D<F32>&) { unreachable("Axis::updatePixelSize"); }
Result Axis::updateTickLabels(const std::vector<std::string>&, const std::vector<std::string>&) { unreachable("Axis::updateTickLabels"); }
Result Axis::updateTitles(const std::string&, const std::string&) { unreachable("Axis::updateTitles"); }
const Extent2D<F32>& Axis::paddingScale() const { unreachable("Axis::paddingScale"); }

}  // namespace Components
}  // namespace Jetstream::Render
