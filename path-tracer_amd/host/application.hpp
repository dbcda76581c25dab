// application.hpp -- headless stand-in for include/application.hpp:9-26: the reference opens an SDL
// window and renders until it is closed; this one parses a command line, renders a fixed number of
// frames and writes the image and a JSON line of metrics.  With --gpus N it drives N renderers from this one
// host thread (the library's calls are asynchronous), frames dealt round-robin, and sums their float
// images with pt_accum_allreduce before the image is resolved.
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
    uint32_t _gpus = 1;              // renderers, one per GPU: renderer r traces frames r, r + N, ...
    bool _rehearse = false;          // all renderers on --device (a box with one GPU)
    std::string _output, _dumpResult, _dumpAccum;
    std::vector<std::shared_ptr<HipContext>> _contexts;
    std::vector<std::unique_ptr<Renderer>> _renderers;
};
