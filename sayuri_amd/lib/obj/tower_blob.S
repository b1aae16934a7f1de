	.section .rodata
	.globl sayuri_tower_hsaco
	.type sayuri_tower_hsaco,@object
	.balign 4096
sayuri_tower_hsaco:
	.incbin "/root/repo/sayuri_amd/lib/obj/tower.hsaco"
	.size sayuri_tower_hsaco, .-sayuri_tower_hsaco
	.section .note.GNU-stack,"",@progbits
