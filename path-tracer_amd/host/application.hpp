// application.hpp -- headless stand-in for include/application.hpp:9-26: the reference opens an SDL
// window and renders until it is closed; this one parses a command line, renders a fixed number of
// frames and writes the image and a JSON line of metrics.
#pragma once
#include "renderer.hpp"

class Application {
public:
    Application(int argc, char** argv);
    int Run();

private:
    RendererInitInfo _info;
    uint32_t _frames = 1;
    int _device = 0;
    std::string _output, _dumpResult;
    std::shared_ptr<HipContext> _context;
    std::unique_ptr<Renderer> _renderer;
};
