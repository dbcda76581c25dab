// main.cpp -- source/main.cpp:3-7 of the reference: construct the application, run it.
#include "application.hpp"

int main(int argc, char** argv)
{
    Application app(argc, argv);
    return app.Run();
}
