/* stub: the reference's scene.cpp includes <GL/glew.h> without using it (scene.cpp:25) */
