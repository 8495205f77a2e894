/* fabber_main.cc - `fabber` executable (the reference's fabber_main.cc) */
#include "fabber_core/fabber_core.h"

int main(int argc, char **argv)
{
    return execute(argc, argv);
}
